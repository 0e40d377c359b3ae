// Row-wise / gather / scatter kernels of the page-read path (gfx950 / MI355X).  All HBM-bound: 16-byte
// vector accesses, one wave per row for the norms, LDS tiles wherever a transpose is needed.
// Every kernel cites the reference-side operation it replaces (HF = transformers, the library the
// reference's run_ocr calls at ocr_agent/tools.py:756-769).
#include "common.h"
#include "hwocr.h"

namespace {

constexpr int MAXC = 8;  // 16-byte chunks per lane per row -> row width <= 64*8*8 = 4096

// ------------------------------------------------------------------------------------------------
// patchify: resized uint8 page -> bf16 patch rows.  Replaces rescale + normalize + patchify of
// HF image_processing_pil_qwen2_vl.py:152-187,:226-229 followed by the .to(bf16) of PatchEmbed
// (modeling_qwen2_vl.py:266-274).  (v/255 - mean)/std for the 256 pixel values x 3 channels is a host-built
// table (same numpy arithmetic as the reference library), so the kernel is an exact gather.
// Row order: (gh/m, gw/m, m, m); column order: (C, T, ph, pw), T = temporal duplicate.
// ------------------------------------------------------------------------------------------------
struct PatchifyArgs {
  const uint8_t* img; const bf16* lut; bf16* out; const int* row_src;
  int nimg, H, W, gh, gw, patch, merge, tps, kreal, kpad, img_ld;
};
__global__ __launch_bounds__(256) void patchify_kernel(PatchifyArgs a) {
  const int chunks = a.kpad >> 3;
  const long total = (long)a.nimg * a.gh * a.gw * chunks;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = gid % chunks;
  const long row = gid / chunks;
  const int P = a.gh * a.gw;
  const int im = row / P, p = row % P;
  const int mm = a.merge * a.merge;
  const int ps = a.row_src ? a.row_src[p] : p;  // output row p shows source patch ps (window order, Qwen2.5-VL)
  const int blk = ps / mm, within = ps % mm;
  const int bw_n = a.gw / a.merge;
  const int pr = (blk / bw_n) * a.merge + within / a.merge;
  const int pc = (blk % bw_n) * a.merge + within % a.merge;
  const int pp = a.patch * a.patch;
  const uint8_t* base = a.img + (long)im * a.H * a.W * 3;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = ch * 8 + e;
    bf16 v = (bf16)0.0f;
    if (k < a.kreal) {
      const int c = k / (a.tps * pp);
      const int rem = k % pp;
      const int py = rem / a.patch, px = rem % a.patch;
      const uint8_t pix = base[((long)(pr * a.patch + py) * a.W + (pc * a.patch + px)) * 3 + c];
      v = a.lut[c * 256 + pix];
    }
    o[e] = v;
  }
  *(bf16x8*)(a.out + ((long)im * a.img_ld + p) * a.kpad + ch * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (vision tower: HF modeling_qwen2_vl.py:428-429, merger ln_q :281).  fp32 statistics,
// one rounding at the end, as nn.LayerNorm does on a bf16 tensor.
// ------------------------------------------------------------------------------------------------
// E4M3 row emission shared by the row quantiser and the norms that feed an fp8 GEMM: the wave holds one row as bf16
// chunks o[i] (chunk lane + 64 i); scale = max|o| / 448, codes = e4m3(o * (448 / max|o|)), round-to-nearest-even.
__device__ __forceinline__ int2 e4m3_pack8(const bf16x8& v, float inv) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(bf2f(v[e]) * inv, -448.0f), 448.0f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return make_int2(lo, hi);
}
template <int NC>
__device__ __forceinline__ void e4m3_emit_row(const bf16x8 (&o)[NC], int lane, int nch, unsigned char* dst, float* scale) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(o[i][e])));
  amax = wave_max(amax);
  const float inv = amax > 0.f ? 448.0f / amax : 0.f;
  if (lane == 0) *scale = amax > 0.f ? amax / 448.0f : 1.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < nch) *(int2*)(dst + (lane + 64 * i) * 8) = e4m3_pack8(o[i], inv);
}

struct LayerNormArgs {
  const bf16* x; const bf16* w; const bf16* b; bf16* out; int rows, D, ldx, ldo; float eps;
  unsigned char* q8; float* q8s; int ldq;  // q8 != NULL: the bf16 row is not stored, its E4M3 codes + scale are
};
// NC = 16-byte chunks per lane (row width <= 512 NC): sized to the row, the registers of an 8-chunk instance (184 VGPRs, 2 waves per
// SIMD) left too few rows in flight per CU to cover the memory latency - 2.0 TB/s at the tower's 1280-wide rows
template <int NC>
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = a.D >> 3;
  // weight and bias of this lane's chunks: the same for every row of the loop
  bf16x8 g[NC], bb[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = min(lane + 64 * i, nch - 1);
    g[i] = *(const bf16x8*)(a.w + ch * 8);
    bb[i] = *(const bf16x8*)(a.b + ch * 8);
  }
  auto load_row = [&](int row, bf16x8 (&v)[NC]) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) v[i] = *(const bf16x8*)(a.x + (long)row * a.ldx + ch * 8);
    }
  };
  // grid-stride over rows (HWOCR_LN_GRID workgroups, default 4096), one wave per row; the NEXT row of the wave is requested before
  // this one is reduced (a row is one round trip to HBM followed by two wave reductions: with nothing in flight behind it the
  // wave idles for the whole trip - 69 -> 60 us per 62208 x 1280 launch)
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + w;
  bf16x8 cur[NC], nxt[NC];
  if (row < a.rows) load_row(row, cur);
  for (; row < a.rows; row += stride) {
    if (row + stride < a.rows) load_row(row + stride, nxt);
    float x[NC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (lane + 64 * i < nch)
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[i][e] = bf2f(cur[i][e]); s += x[i][e]; }
    const float mean = wave_sum(s) / a.D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (lane + 64 * i < nch)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = x[i][e] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / a.D + a.eps);
    bf16x8 o[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[i][e] = f2bf((x[i][e] - mean) * rstd * bf2f(g[i][e]) + bf2f(bb[i][e]));
        if (!a.q8) *(bf16x8*)(a.out + (long)row * a.ldo + ch * 8) = o[i];
      }
    }
    if (a.q8) e4m3_emit_row(o, lane, nch, a.q8 + (long)row * a.ldq, a.q8s + row);
#pragma unroll
    for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
  }
}
// ------------------------------------------------------------------------------------------------
// (split-K combine + bias + residual) + RMSNorm.  Replaces Qwen2VLRMSNorm (HF modeling_qwen2_vl.py:96-110:
// fp32 x*rsqrt(mean(x^2)+eps) -> cast to bf16 -> weight * that) and the two residual adds of the decoder
// layer (:591-600).  With slabs: h <- bf16(bf16(sum slabs + bias) + h) is written back first.
// row_index (optional) gathers source rows (last prompt token of every read before the LM head).
// ------------------------------------------------------------------------------------------------
struct RmsArgs {
  const float* slabs; int nslab; long slab_stride; int ld_slab;
  const bf16* bias; bf16* h; int ldh;
  const bf16* w; bf16* out; int ldo;
  const int* row_index; int rows, D; float eps; int gemma;
  unsigned char* q8 = nullptr; float* q8s = nullptr; int ldq = 0;  // add_rmsnorm_kernel only: E4M3 codes + scale instead of out
};
template <int NC>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(RmsArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= a.rows) return;
  const int src = a.row_index ? a.row_index[row] : row;
  const int nch = a.D >> 3;
  float x[NC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const bf16x8 hv = *(const bf16x8*)(a.h + (long)src * a.ldh + ch * 8);
      if (a.nslab > 0) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = 0.f;
        for (int s = 0; s < a.nslab; ++s) {
          const float* p = a.slabs + s * a.slab_stride + (long)src * a.ld_slab + ch * 8;
          const f32x4 p0 = *(const f32x4*)p, p1 = *(const f32x4*)(p + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] += p0[e]; y[4 + e] += p1[e]; }
        }
        if (a.bias) {
          const bf16x8 bb = *(const bf16x8*)(a.bias + ch * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] += bf2f(bb[e]);
        }
        bf16x8 hn;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hn[e] = f2bf(rbf(y[e]) + bf2f(hv[e])); x[i][e] = bf2f(hn[e]); }
        *(bf16x8*)(a.h + (long)src * a.ldh + ch * 8) = hn;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[i][e] = bf2f(hv[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[i][e] * x[i][e];
    }
  }
  if (!a.out && !a.q8) return;
  const float rstd = rsqrtf(wave_sum(ss) / a.D + a.eps);
  bf16x8 o[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const bf16x8 g = *(const bf16x8*)(a.w + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[i][e] = a.gemma ? f2bf(x[i][e] * rstd * (1.0f + bf2f(g[e]))) : f2bf(bf2f(g[e]) * rbf(x[i][e] * rstd));
      if (!a.q8) *(bf16x8*)(a.out + (long)row * a.ldo + ch * 8) = o[i];
    }
  }
  if (a.q8) e4m3_emit_row(o, lane, nch, a.q8 + (long)row * a.ldq, a.q8s + row);
}

// Few rows (decode: one row per read): one workgroup of NT threads per row (256: D <= 2048, 512: D <= 4096), every
// thread owns <= 1 chunk of 8 elements, all slab loads of a thread are independent and in flight together; the row
// statistic goes through LDS.
template <int NT>
__global__ __launch_bounds__(NT) void add_rmsnorm_row_kernel(RmsArgs a) {
  __shared__ float s_part[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row = blockIdx.x;
  const int src = a.row_index ? a.row_index[row] : row;
  const int nch = a.D >> 3;
  const bool live = tid < nch;
  float x[8];
  float ss = 0.f;
  if (live) {
    const bf16x8 hv = *(const bf16x8*)(a.h + (long)src * a.ldh + tid * 8);
    if (a.nslab > 0) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = 0.f;
      const float* p = a.slabs + (long)src * a.ld_slab + tid * 8;
      int s = 0;
      for (; s + 4 <= a.nslab; s += 4) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[2 * u] = *(const f32x4*)(p + (s + u) * a.slab_stride);
          v[2 * u + 1] = *(const f32x4*)(p + (s + u) * a.slab_stride + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { y[e] += v[2 * u][e]; y[4 + e] += v[2 * u + 1][e]; }
      }
      for (; s < a.nslab; ++s) {
        const f32x4 p0 = *(const f32x4*)(p + s * a.slab_stride), p1 = *(const f32x4*)(p + s * a.slab_stride + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] += p0[e]; y[4 + e] += p1[e]; }
      }
      if (a.bias) {
        const bf16x8 bb = *(const bf16x8*)(a.bias + tid * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += bf2f(bb[e]);
      }
      bf16x8 hn;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hn[e] = f2bf(rbf(y[e]) + bf2f(hv[e])); x[e] = bf2f(hn[e]); }
      *(bf16x8*)(a.h + (long)src * a.ldh + tid * 8) = hn;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = bf2f(hv[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
  }
  if (!a.out) return;
  ss = wave_sum(ss);
  if (lane == 0) s_part[w] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) tot += s_part[i];
  const float rstd = rsqrtf(tot / a.D + a.eps);
  if (live) {
    const bf16x8 g = *(const bf16x8*)(a.w + tid * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = a.gemma ? f2bf(x[e] * rstd * (1.0f + bf2f(g[e]))) : f2bf(bf2f(g[e]) * rbf(x[e] * rstd));
    *(bf16x8*)(a.out + (long)row * a.ldo + tid * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Vision rope + head split.  qkv[token][3][head][hd] -> Q,K [head][token][hd] (rotated, fp32 math then one
// rounding: HF apply_rotary_pos_emb_vision modeling_qwen2_vl.py:239-248) and V^T [head][hd][token].
// ------------------------------------------------------------------------------------------------
struct VitRopeArgs {
  const bf16* qkv; bf16* Q; bf16* K; bf16* VT;
  const int* pos_h; const int* pos_w; const float* cos_tab; const float* sin_tab;
  int tokens, tok_ld, heads, hd; long q_head_stride, vt_head_stride;
  int interleaved;  // q / k features of a head arrive as rotary pairs side by side: [d0, d0 + hd/2, d1, d1 + hd/2, ...]
};
__global__ __launch_bounds__(256) void vit_rope_split_kernel(VitRopeArgs a) {
  __shared__ bf16 s_v[128][72];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, tid = threadIdx.x;
  const int hd = a.hd, half = hd >> 1, quarter = hd >> 2;
  const int D = a.heads * hd;
  const int per_tok = hd >> 4;  // (d, d+half) chunk pairs per token
  for (int id = tid; id < 64 * per_tok * 2; id += 256) {
    const int which = id / (64 * per_tok);  // 0 = q, 1 = k
    const int rem = id % (64 * per_tok);
    const int tt = rem / per_tok, j = rem % per_tok;
    const int tok = t0 + tt;
    if (tok >= a.tokens) continue;
    bf16x8 va, vb;  // x1 = features 8j .. 8j+7 of the head, x2 = their partners hd/2 further on
    if (a.interleaved) {
      const bf16* src = a.qkv + (long)tok * 3 * D + which * D + h * hd + 16 * j;
      const bf16x8 c0 = *(const bf16x8*)src, c1 = *(const bf16x8*)(src + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = c0[2 * e]; vb[e] = c0[2 * e + 1];
        va[4 + e] = c1[2 * e]; vb[4 + e] = c1[2 * e + 1];
      }
    } else {
      const bf16* src = a.qkv + (long)tok * 3 * D + which * D + h * hd + 8 * j;
      va = *(const bf16x8*)src;
      vb = *(const bf16x8*)(src + half);
    }
    const int ph = a.pos_h[tok], pw = a.pos_w[tok];
    bf16x8 oa, ob;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = 8 * j + e;
      const int p = d < quarter ? ph : pw;
      const int f = d < quarter ? d : d - quarter;
      const float cs = a.cos_tab[p * quarter + f], sn = a.sin_tab[p * quarter + f];
      const float x1 = bf2f(va[e]), x2 = bf2f(vb[e]);
      oa[e] = f2bf(__fadd_rn(__fmul_rn(x1, cs), __fmul_rn(-x2, sn)));
      ob[e] = f2bf(__fadd_rn(__fmul_rn(x2, cs), __fmul_rn(x1, sn)));
    }
    bf16* dst = (which ? a.K : a.Q) + h * a.q_head_stride + (long)tok * hd + 8 * j;
    *(bf16x8*)dst = oa;
    *(bf16x8*)(dst + half) = ob;
  }
  // V: [64 tokens][hd] -> LDS transposed -> V^T rows of 64 tokens (zeros past the last token)
  const int vch = hd >> 3;
  for (int id = tid; id < 64 * vch; id += 256) {
    const int tt = id / vch, j = id % vch;
    const int tok = t0 + tt;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
    if (tok < a.tokens) v = *(const bf16x8*)(a.qkv + (long)tok * 3 * D + 2 * D + h * hd + 8 * j);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_v[8 * j + e][tt] = v[e];
  }
  __syncthreads();
  for (int id = tid; id < hd * 8; id += 256) {
    const int d = id >> 3, c8 = id & 7;
    if (t0 + c8 * 8 < a.tok_ld) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s_v[d][c8 * 8 + e];
      *(bf16x8*)(a.VT + h * a.vt_head_stride + (long)d * a.tok_ld + t0 + c8 * 8) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Decoder M-RoPE + KV-cache write (prefill).  Replaces apply_multimodal_rotary_pos_emb
// (HF modeling_qwen2_vl.py:196-222: bf16 cos/sin, every product and the sum rounded to bf16) and
// DynamicLayer.update (HF cache_utils.py:127-145: torch.cat realloc) with an in-place append into
// K[seq][kvh][slot][128] and V^T[seq][kvh][128][slot].
// ------------------------------------------------------------------------------------------------
struct MropeArgs {
  const bf16* qkv; bf16* Q; bf16* K; bf16* VT;
  const int* pos;  // pos[3][rows]
  const bf16* cos_tab; const bf16* sin_tab;                  // [maxpos][64]
  int rows, rows_per_seq, Hq, Hkv, sec0, sec1;
  long k_seq, k_head, v_seq, v_head, v_row;
  int tiled;
};
// DHD: decoder head_dim, 128 (Qwen families) or 256 (Gemma).  The rope tables are [maxpos][DHD/2]; with sec0 >= DHD/2 every
// frequency takes the first position axis, i.e. plain 1-D RoPE (HF gemma/modeling_gemma.py:166-190: the same bf16 chain).
template <int DHD>
__device__ __forceinline__ void mrope_pair(const bf16x8 va, const bf16x8 vb, int d0, const int pt, const int ph,
                                           const int pw, const MropeArgs& a, bf16x8& oa, bf16x8& ob) {
  constexpr int HALF = DHD / 2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = d0 + e;  // 0..HALF-1
    const int p = i < a.sec0 ? pt : (i < a.sec1 ? ph : pw);
    const float cs = bf2f(a.cos_tab[p * HALF + i]), sn = bf2f(a.sin_tab[p * HALF + i]);
    const float x1 = bf2f(va[e]), x2 = bf2f(vb[e]);
    oa[e] = f2bf(rbf(x1 * cs) + rbf(-x2 * sn));
    ob[e] = f2bf(rbf(x2 * cs) + rbf(x1 * sn));
  }
}

template <int DHD>
__global__ __launch_bounds__(256) void mrope_kv_prefill_kernel(MropeArgs a) {
  constexpr int HALF = DHD / 2, HC = HALF / 8;  // 16-byte chunks per half head
  __shared__ bf16 s_v[64][DHD + 2];
  const int r0 = blockIdx.x * 64, hy = blockIdx.y, tid = threadIdx.x;
  const int W = (a.Hq + 2 * a.Hkv) * DHD;
  if (hy < a.Hq + a.Hkv) {
    const bool isk = hy >= a.Hq;
    const int col0 = isk ? a.Hq * DHD + (hy - a.Hq) * DHD : hy * DHD;
    for (int id = tid; id < 64 * HC; id += 256) {
      const int rr = id / HC, j = id % HC;
      const int row = r0 + rr;
      if (row >= a.rows) continue;
      const bf16* src = a.qkv + (long)row * W + col0 + 8 * j;
      bf16x8 oa, ob;
      mrope_pair<DHD>(*(const bf16x8*)src, *(const bf16x8*)(src + HALF), 8 * j, a.pos[row], a.pos[a.rows + row],
                      a.pos[2 * a.rows + row], a, oa, ob);
      const int sq = row / a.rows_per_seq, slot = row % a.rows_per_seq;
      if (isk) {
        bf16* kb = a.K + sq * a.k_seq + (hy - a.Hq) * a.k_head;
        *(bf16x8*)(kb + (a.tiled ? kv_tiled_k(slot, 8 * j) : (long)slot * DHD + 8 * j)) = oa;
        *(bf16x8*)(kb + (a.tiled ? kv_tiled_k(slot, HALF + 8 * j) : (long)slot * DHD + HALF + 8 * j)) = ob;
      } else {
        bf16* dst = a.Q + (long)row * a.Hq * DHD + hy * DHD + 8 * j;
        *(bf16x8*)dst = oa;
        *(bf16x8*)(dst + HALF) = ob;
      }
    }
  } else {
    const int hk = hy - a.Hq - a.Hkv;
    for (int id = tid; id < 64 * (DHD / 8); id += 256) {
      const int rr = id / (DHD / 8), j = id % (DHD / 8);
      const int row = r0 + rr;
      if (row >= a.rows) continue;
      const bf16x8 v = *(const bf16x8*)(a.qkv + (long)row * W + (a.Hq + a.Hkv + hk) * DHD + 8 * j);
#pragma unroll
      for (int e = 0; e < 8; ++e) s_v[rr][8 * j + e] = v[e];
    }
    __syncthreads();
    // lane <-> row: consecutive lanes hit consecutive cache slots, so each wave store is one 128-byte line
    const int rr = tid & 63, dg = tid >> 6;
    const int row = r0 + rr;
    if (row < a.rows) {
      const int slot = row % a.rows_per_seq;
      bf16* vb = a.VT + (row / a.rows_per_seq) * a.v_seq + hk * a.v_head;
      for (int d = dg * (DHD / 4); d < (dg + 1) * (DHD / 4); ++d)
        vb[a.tiled ? kv_tiled_v(d, slot) : (long)d * a.v_row + slot] = s_v[rr][d];
    }
  }
}

// Decode variant: one row per read, input = split-K slabs of the fused QKV projection (+bias).
struct DecQkvArgs {
  const float* slabs; int nslab; long slab_stride;
  const bf16* bias; bf16* Q; bf16* K; bf16* VT;
  const int* lens; const int* rope_delta;
  const bf16* cos_tab; const bf16* sin_tab;
  int Hq, Hkv; long k_seq, k_head, v_seq, v_head, v_row; int tiled;
  int ctx, max_pos; int* status;
};
template <int DHD>
__global__ __launch_bounds__(256) void decode_qkv_finish_kernel(DecQkvArgs a) {
  constexpr int HALF = DHD / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* row = (bf16*)smem;  // [(Hq+2Hkv)*DHD]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int W = (a.Hq + 2 * a.Hkv) * DHD;
  for (int ch = tid; ch < W / 8; ch += 256) {
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = 0.f;
    for (int s = 0; s < a.nslab; ++s) {
      const float* p = a.slabs + s * a.slab_stride + (long)b * W + ch * 8;
      const f32x4 p0 = *(const f32x4*)p, p1 = *(const f32x4*)(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] += p0[e]; y[4 + e] += p1[e]; }
    }
    if (a.bias) {
      const bf16x8 bb = *(const bf16x8*)(a.bias + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] += bf2f(bb[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) row[ch * 8 + e] = f2bf(y[e]);
  }
  __syncthreads();
  // Invariants the host keeps (engine.py): 1 <= lens <= ctx and 0 <= lens - 1 + rope_delta < max_pos (all position axes
  // coincide on generated tokens).  A read that breaks them is NOT repaired here: its row is skipped — no cache write, no
  // table read outside the allocation — and bit 0 of *status is raised for the host to turn into an error.
  const int slot = a.lens[b] - 1;
  const int p = slot + a.rope_delta[b];
  if (slot < 0 || slot >= a.ctx || p < 0 || p >= a.max_pos) {  // uniform over the workgroup
    if (tid == 0 && a.status) atomicOr(a.status, HWOCR_STATUS_BAD_POSITION);
    return;
  }
  // q and k heads: pairs (d, d + DHD/2)
  for (int id = tid; id < (a.Hq + a.Hkv) * HALF; id += 256) {
    const int hy = id / HALF, i = id % HALF;
    const float cs = bf2f(a.cos_tab[p * HALF + i]), sn = bf2f(a.sin_tab[p * HALF + i]);
    const float x1 = bf2f(row[hy * DHD + i]), x2 = bf2f(row[hy * DHD + HALF + i]);
    const bf16 oa = f2bf(rbf(x1 * cs) + rbf(-x2 * sn));
    const bf16 ob = f2bf(rbf(x2 * cs) + rbf(x1 * sn));
    if (hy < a.Hq) {
      bf16* dst = a.Q + ((long)b * a.Hq + hy) * DHD;
      dst[i] = oa;
      dst[HALF + i] = ob;
    } else {
      bf16* kb = a.K + b * a.k_seq + (hy - a.Hq) * a.k_head;
      kb[a.tiled ? kv_tiled_k(slot, i) : (long)slot * DHD + i] = oa;
      kb[a.tiled ? kv_tiled_k(slot, HALF + i) : (long)slot * DHD + HALF + i] = ob;
    }
  }
  for (int id = tid; id < a.Hkv * DHD; id += 256) {
    const int hk = id / DHD, d = id % DHD;
    a.VT[b * a.v_seq + hk * a.v_head + (a.tiled ? kv_tiled_v(d, slot) : (long)d * a.v_row + slot)] =
        row[(a.Hq + a.Hkv) * DHD + id];
  }
}

// ------------------------------------------------------------------------------------------------
// Token embedding gather + image-token splice (HF modeling_qwen2_vl.py get_placeholder_mask +
// masked_scatter: image rows replace the <|image_pad|> embeddings).
// ------------------------------------------------------------------------------------------------
struct EmbedArgs { const int* ids; const int* img_row; const bf16* table; const bf16* img; bf16* out; int rows, D; float scale; };
__global__ __launch_bounds__(256) void embed_splice_kernel(EmbedArgs a) {
  const int chunks = a.D >> 3;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)a.rows * chunks) return;
  const int row = gid / chunks, ch = gid % chunks;
  const int ir = a.img_row ? a.img_row[row] : -1;
  bf16x8 v = ir >= 0 ? *(const bf16x8*)(a.img + (long)ir * a.D + ch * 8)
                     : *(const bf16x8*)(a.table + (long)a.ids[row] * a.D + ch * 8);
  if (a.scale != 1.0f && ir < 0)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) * rbf(a.scale));
  *(bf16x8*)(a.out + (long)row * a.D + ch * 8) = v;
}

// ------------------------------------------------------------------------------------------------
// Greedy token select + stop bookkeeping, all on device so a decode step never syncs with the host.
// Replaces HF generation/utils.py:2894-2937: logits[:, -1].float() -> (min-new-tokens EOS suppression) ->
// argmax -> pad finished rows -> append -> EOS stop flags.
// ------------------------------------------------------------------------------------------------
struct SelectArgs {
  const bf16* logits; int ldl, V;
  int* cur_ids; int* lens; int* n_gen; int* finished; int* out_tokens; int max_new, min_new;
  int eos[4]; int n_eos; int pad_id;
  unsigned* seen; int seen_ld; float rep_penalty;  // bitmap [nseq][seen_ld words] of ids in prompt + output so far, or NULL
  int* split_ws = nullptr;                         // argmax_advance_kernel<., PARTS > 1>
};
// One workgroup of 16 waves per read: a row is V x 2 B (300 KB at V = 151936) and only `nseq` CUs take part, so what
// matters is loads in flight per CU — 4 independent 16-byte loads per thread per trip (256 threads, one load per trip:
// 95 us per step at 126 reads).
// PARTS > 1 (few reads in flight: one workgroup per read leaves the chip empty and took 37 us of a 3-read step's 1480): the row is
// cut into PARTS contiguous ranges, one workgroup of NT threads each; every workgroup leaves its (maximum, index) in a.split_ws and
// the one that arrives last picks the winner and does the bookkeeping.  (value, lower index) is a total order, so the pick does not
// depend on how the row was cut: the same token as the one-workgroup form.
constexpr int ARG_THREADS = 1024;
constexpr int SEL_PARTS = 16, SEL_WS_INTS = 40;  // split_ws per read: [0] arrivals, [1 + 2 part] = (value bits, index); hwocr.h
template <int NT, int PARTS>
__global__ __launch_bounds__(NT) void argmax_advance_kernel(SelectArgs a) {
  constexpr int NW = NT / 64;
  __shared__ float s_val[NW];
  __shared__ int s_idx[NW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bf16* row = a.logits + (long)b * a.ldl;
  const bool suppress = a.n_gen[b] < a.min_new;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int nch = a.V / 8;
  // RepetitionPenaltyLogitsProcessor (HF generation/logits_process.py: score < 0 ? score * p : score / p for every id
  // already in input_ids), on the fp32 copy of the logits like every processor
  // The token fed to produce these logits (cur_ids: the previous pick, or the teacher-forced one the host wrote) joins the
  // bitmap first; the prompt's ids were put there by the host, and the call that follows a prefill has n_gen == 0.
  // (PARTS > 1: the workgroups of a read treat the fed token as seen while they scan; the finishing one records it.)
  int fed = -1;
  if (a.seen && a.n_gen[b] > 0 && !a.finished[b]) {
    const int t = a.cur_ids[b];
    if (t >= 0 && t < a.V) fed = t;
  }
  if (PARTS == 1 && a.seen) {
    if (tid == 0 && fed >= 0) a.seen[(long)b * a.seen_ld + (fed >> 5)] |= 1u << (fed & 31);
    __syncthreads();
  }
  const unsigned* seen = a.seen ? a.seen + (long)b * a.seen_ld : nullptr;
  auto take = [&](const bf16x8& v, int ch) {
    unsigned bits = seen ? (seen[ch >> 2] >> ((ch & 3) * 8)) & 0xffu : 0u;  // the 8 ids of this chunk
    if (PARTS > 1 && (fed >> 3) == ch) bits |= 1u << (fed & 7);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = bf2f(v[e]);
      const int idx = ch * 8 + e;
      if ((bits >> e) & 1u) x = x < 0.f ? x * a.rep_penalty : x / a.rep_penalty;
      if (suppress)
        for (int k = 0; k < a.n_eos; ++k)
          if (idx == a.eos[k]) x = -INFINITY;
      if (x > best || (x == best && idx < bi)) { best = x; bi = idx; }
    }
  };
  const int per = (nch + PARTS - 1) / PARTS;
  const int c0 = PARTS > 1 ? (int)blockIdx.y * per : 0, c1 = PARTS > 1 ? min(nch, c0 + per) : nch;
  int ch = c0 + tid;
  for (; ch + 3 * NT < c1; ch += 4 * NT) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(row + (long)(ch + u * NT) * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) take(v[u], ch + u * NT);
  }
  for (; ch < c1; ch += NT) take(*(const bf16x8*)(row + (long)ch * 8), ch);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_val[w] = best; s_idx[w] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < NW; ++k)
      if (s_val[k] > best || (s_val[k] == best && s_idx[k] < bi)) { best = s_val[k]; bi = s_idx[k]; }
    if (PARTS > 1) {
      // device-scope (sc1) stores / loads around a device-scope count, as in attn_decode_kernel: the other parts ran on other XCDs
      int* ws = a.split_ws + (long)b * SEL_WS_INTS;
      __hip_atomic_store(ws + 1 + 2 * blockIdx.y, __float_as_int(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ws + 2 + 2 * blockIdx.y, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (__hip_atomic_fetch_add(ws, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != PARTS - 1) return;
      int pv[PARTS], pi[PARTS];
#pragma unroll
      for (int k = 0; k < PARTS; ++k) {
        pv[k] = __hip_atomic_load(ws + 1 + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pi[k] = __hip_atomic_load(ws + 2 + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      best = -INFINITY;
      bi = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < PARTS; ++k) {
        const float v = __int_as_float(pv[k]);
        if (v > best || (v == best && pi[k] < bi)) { best = v; bi = pi[k]; }
      }
      __hip_atomic_store(ws, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (fed >= 0) a.seen[(long)b * a.seen_ld + (fed >> 5)] |= 1u << (fed & 31);
    }
    int tok = bi;
    if (a.finished[b]) {
      tok = a.pad_id;
    } else {
      for (int k = 0; k < a.n_eos; ++k)
        if (tok == a.eos[k]) a.finished[b] = 1;
    }
    const int n = a.n_gen[b];
    if (n < a.max_new) a.out_tokens[(long)b * a.max_new + n] = tok;
    a.n_gen[b] = n + 1;
    a.cur_ids[b] = tok;
    a.lens[b] += 1;
  }
}


// ------------------------------------------------------------------------------------------------
// Token draw of generate(do_sample=True): temperature -> top-k -> top-p -> multinomial (HF generation/logits_process.py warpers +
// generation/utils.py _sample), as the exact integer procedure "hwocr sampling v1" that oracle/sampling.py spells out: distance keys
// q = floor((max - score) * 65536) (clamped), fixed-point weights floor(2^32 * 2^(-d * log2e / T)) from a non-fused fp32 polynomial,
// radix selection of the top-k / nucleus thresholds over (count | mass) histograms, Philox4x32-10 keyed by (seed; read, step), inverse
// CDF in token-id order.  Integer sums and a counter-based RNG: a read's tokens depend on (seed, read index, step) only.
// One workgroup per read; the row (<= 500 KB of bf16 logits) is walked ~6 times out of L2.
// ------------------------------------------------------------------------------------------------
struct SampleArgs {
  SelectArgs g;
  float c;        // log2(e) / temperature, fp32
  int top_k;      // <= 0 or >= V: off
  float top_p;    // >= 1: off
  unsigned seed_lo, seed_hi;
  const int* read_ids;  // the caller's number of each read (RNG counter), or nullptr: the row index
  unsigned long long* dbg;  // optional [nseq][8]: M bits, top-k key, W, P, nucleus key, kept mass, target, token
};
constexpr int SMP_QMAX = (1 << 22) - 1;
constexpr int SMP_BINS = 2048;

__device__ __forceinline__ unsigned long long smp_weight(float d, float c) {
  const float y = -__fmul_rn(d, c);
  if (!(y > -40.0f)) return 0ull;
  const float n = __builtin_rintf(y);
  const float f = __fsub_rn(y, n);
  float r = 0.00015403530393381608f;
  r = __fadd_rn(__fmul_rn(r, f), 0.0013333558146428443f);
  r = __fadd_rn(__fmul_rn(r, f), 0.009618129107628477f);
  r = __fadd_rn(__fmul_rn(r, f), 0.05550410866482158f);
  r = __fadd_rn(__fmul_rn(r, f), 0.2402265069591007f);
  r = __fadd_rn(__fmul_rn(r, f), 0.6931471805599453f);
  r = __fadd_rn(__fmul_rn(r, f), 1.0f);
  return (unsigned long long)__builtin_ldexp((double)r, (int)n + 32);
}
__device__ __forceinline__ int smp_key(float d) {
  const float t = __builtin_floorf(__fmul_rn(d, 65536.0f));
  return (t >= (float)SMP_QMAX || !(t == t) || __builtin_isinf(t)) ? SMP_QMAX : (int)t;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = shfl_u64(v, max(lane - o, 0));
    if (lane >= o) v += u;
  }
  return v;
}

__global__ __launch_bounds__(ARG_THREADS) void sample_advance_kernel(SampleArgs a) {
  constexpr int NW = ARG_THREADS / 64;
  __shared__ unsigned long long s_hist[SMP_BINS];
  __shared__ unsigned long long s_wave[NW];
  __shared__ float s_fmax[NW];
  __shared__ unsigned long long s_res[2];  // crossing bin, mass before it
  __shared__ int s_tok;
  const SelectArgs& g = a.g;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bf16* row = g.logits + (long)b * g.ldl;
  const bool suppress = g.n_gen[b] < g.min_new;
  const int nch = g.V / 8;
  if (g.seen) {  // the token that produced these logits joins the repetition bitmap first (as argmax_advance_kernel)
    if (tid == 0 && g.n_gen[b] > 0 && !g.finished[b]) {
      const int t = g.cur_ids[b];
      if (t >= 0 && t < g.V) g.seen[(long)b * g.seen_ld + (t >> 5)] |= 1u << (t & 31);
    }
    __syncthreads();
  }
  const unsigned* seen = g.seen ? g.seen + (long)b * g.seen_ld : nullptr;
  // the 8 fp32 scores of chunk ch (ids 8 ch .. 8 ch + 7): repetition penalty, EOS suppression - exactly the greedy path's
  auto scores = [&](int ch, float (&x)[8]) {
    const bf16x8 v = *(const bf16x8*)(row + (long)ch * 8);
    const unsigned bits = seen ? (seen[ch >> 2] >> ((ch & 3) * 8)) & 0xffu : 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = bf2f(v[e]);
      if ((bits >> e) & 1u) s = s < 0.f ? s * g.rep_penalty : s / g.rep_penalty;
      if (suppress)
        for (int k = 0; k < g.n_eos; ++k)
          if (ch * 8 + e == g.eos[k]) s = -INFINITY;
      x[e] = s;
    }
  };
  // ---- the maximum
  float mx = -INFINITY;
  for (int ch = tid; ch < nch; ch += ARG_THREADS) {
    float x[8];
    scores(ch, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, x[e]);
  }
  mx = wave_max(mx);
  if (lane == 0) s_fmax[w] = mx;
  __syncthreads();
  float M = s_fmax[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) M = fmaxf(M, s_fmax[k]);

  // first bin (ascending key order = descending score) at which the running sum of s_hist reaches `need`; all bins if it never does
  auto crossing = [&](unsigned long long need, unsigned long long& before, unsigned long long& total) -> int {
    const unsigned long long h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
    const unsigned long long incl = wave_incl_scan_u64(h0 + h1, lane);
    if (lane == 63) s_wave[w] = incl;
    if (tid == 0) { s_res[0] = SMP_BINS - 1; s_res[1] = ~0ull; }
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      if (k < w) base += s_wave[k];
      tot += s_wave[k];
    }
    const unsigned long long excl = base + incl - (h0 + h1);
    if (excl < need && need <= excl + h0 + h1) {  // exactly one thread
      const bool first = need <= excl + h0;
      s_res[0] = 2 * tid + (first ? 0 : 1);
      s_res[1] = first ? excl : excl + h0;
    }
    __syncthreads();
    const int bin = (int)s_res[0];
    before = s_res[1] == ~0ull ? 0 : s_res[1];  // (never reached: keep everything; `before` is then unused)
    total = tot;
    __syncthreads();
    return bin;
  };
  auto clear_hist = [&]() {
    s_hist[2 * tid] = 0;
    s_hist[2 * tid + 1] = 0;
    __syncthreads();
  };

  // ---- top-k: the k-th smallest key by two 11-bit levels of counting
  int tk = SMP_QMAX;
  if (a.top_k > 0 && a.top_k < g.V) {
    unsigned long long before, total;
    clear_hist();
    unsigned far = 0;  // keys at the clamp (masked / very distant scores) are counted in a register: one hot bin otherwise
    for (int ch = tid; ch < nch; ch += ARG_THREADS) {
      float x[8];
      scores(ch, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = smp_key(__fsub_rn(M, x[e]));
        if (q == SMP_QMAX) ++far;
        else atomicAdd(&s_hist[q >> 11], 1ull);
      }
    }
    if (far) atomicAdd(&s_hist[SMP_BINS - 1], (unsigned long long)far);
    __syncthreads();
    const int b1 = crossing((unsigned long long)a.top_k, before, total);
    const unsigned long long need2 = (unsigned long long)a.top_k - before;
    clear_hist();
    for (int ch = tid; ch < nch; ch += ARG_THREADS) {
      float x[8];
      scores(ch, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = smp_key(__fsub_rn(M, x[e]));
        if ((q >> 11) == b1) atomicAdd(&s_hist[q & 2047], 1ull);
      }
    }
    __syncthreads();
    const int b2 = crossing(need2, before, total);
    tk = (b1 << 11) | b2;
  }
  // ---- top-p: the key at which the mass from the top reaches P = top_p * W
  int tau = SMP_QMAX;
  unsigned long long W = 0, P = 0;
  if (a.top_p < 1.0f) {
    unsigned long long before, total;
    clear_hist();
    for (int ch = tid; ch < nch; ch += ARG_THREADS) {
      float x[8];
      scores(ch, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = __fsub_rn(M, x[e]);
        const int q = smp_key(d);
        if (q <= tk) {
          const unsigned long long wt = smp_weight(d, a.c);
          if (wt) atomicAdd(&s_hist[q >> 11], wt);
        }
      }
    }
    __syncthreads();
    (void)crossing(~0ull - 1, before, W);  // W = the total (the crossing itself is not reached)
    const double pw = __builtin_floor((double)a.top_p * (double)W);
    P = pw < 1.0 ? 1ull : (unsigned long long)pw;
    const int b1 = crossing(P, before, total);
    const unsigned long long need2 = P - before;
    clear_hist();
    for (int ch = tid; ch < nch; ch += ARG_THREADS) {
      float x[8];
      scores(ch, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = __fsub_rn(M, x[e]);
        const int q = smp_key(d);
        if (q <= tk && (q >> 11) == b1) {
          const unsigned long long wt = smp_weight(d, a.c);
          if (wt) atomicAdd(&s_hist[q & 2047], wt);
        }
      }
    }
    __syncthreads();
    const int b2 = crossing(need2, before, total);
    tau = (b1 << 11) | b2;
  }
  const int qlim = min(tk, tau);
  // ---- the draw: kept mass per wave over contiguous id ranges, then the wave that holds the target walks its range again
  const int cpw = (nch + NW - 1) / NW;  // chunks per wave
  const int c0 = w * cpw, c1 = min(nch, c0 + cpw);
  auto chunk_mass = [&](int ch, unsigned long long (&wt)[8]) -> unsigned long long {
    float x[8];
    scores(ch, x);
    unsigned long long sum = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = __fsub_rn(M, x[e]);
      wt[e] = smp_key(d) <= qlim ? smp_weight(d, a.c) : 0ull;
      sum += wt[e];
    }
    return sum;
  };
  unsigned long long mine = 0;
  for (int ch = c0 + lane; ch < c1; ch += 64) {
    unsigned long long wt[8];
    mine += chunk_mass(ch, wt);
  }
  mine = wave_incl_scan_u64(mine, lane);
  if (lane == 63) s_wave[w] = mine;
  if (tid == 0) s_tok = -1;
  __syncthreads();
  unsigned long long Wk = 0, wbase = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    if (k < w) wbase += s_wave[k];
    Wk += s_wave[k];
  }
  unsigned ctr[4] = {(unsigned)(a.read_ids ? a.read_ids[b] : b), (unsigned)g.n_gen[b], 0u, 0u};
  philox4x32_10(ctr, a.seed_lo, a.seed_hi);
  const unsigned long long r64 = ((unsigned long long)ctr[1] << 32) | ctr[0];
  const unsigned long long target = __umul64hi(Wk, r64);
  if (wbase <= target && target < wbase + s_wave[w]) {  // this wave's id range holds the target (wave-uniform)
    unsigned long long run = wbase;
    for (int cb = c0; cb < c1; cb += 64) {
      const int ch = cb + lane;
      unsigned long long wt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const unsigned long long m = ch < c1 ? chunk_mass(ch, wt) : 0ull;
      const unsigned long long incl = wave_incl_scan_u64(m, lane);
      const unsigned long long trip = shfl_u64(incl, 63);
      if (run + trip > target) {
        const unsigned long long hit = __ballot(run + incl > target);
        const int first = __ffsll((long long)hit) - 1;
        if (lane == first) {
          unsigned long long acc = run + incl - m;
          int tok = ch * 8 + 7;
          for (int e = 0; e < 8; ++e) {
            acc += wt[e];
            if (acc > target) { tok = ch * 8 + e; break; }
          }
          s_tok = tok;
        }
        break;
      }
      run += trip;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int tok = s_tok;
    if (a.dbg) {
      unsigned long long* o = a.dbg + (long)b * 8;
      o[0] = (unsigned long long)__float_as_uint(M);
      o[1] = (unsigned long long)tk;
      o[2] = W;
      o[3] = P;
      o[4] = (unsigned long long)tau;
      o[5] = Wk;
      o[6] = target;
      o[7] = (unsigned long long)(long long)tok;
    }
    if (g.finished[b]) {
      tok = g.pad_id;
    } else {
      for (int k = 0; k < g.n_eos; ++k)
        if (tok == g.eos[k]) g.finished[b] = 1;
    }
    const int n = g.n_gen[b];
    if (n < g.max_new) g.out_tokens[(long)b * g.max_new + n] = tok;
    g.n_gen[b] = n + 1;
    g.cur_ids[b] = tok;
    g.lens[b] += 1;
  }
}

}  // namespace

extern "C" int hwocr_patchify(const void* img, const void* lut, void* out, int nimg, int H, int W, int patch,
                              int merge, int tps, int kpad, int rows_per_img_ld, const int* row_src,
                              hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (nimg <= 0 || H % (patch * merge) || W % (patch * merge) || kpad % 8 || kpad < 3 * tps * patch * patch ||
      rows_per_img_ld < (H / patch) * (W / patch))
    return HWOCR_EINVAL;
  PatchifyArgs a{(const uint8_t*)img, (const bf16*)lut, (bf16*)out, row_src, nimg, H, W, H / patch, W / patch,
                 patch, merge, tps, 3 * tps * patch * patch, kpad, rows_per_img_ld};
  const long total = (long)nimg * a.gh * a.gw * (kpad / 8);
  HWOCR_PLAN("patchify_kernel nimg=%d H=%d W=%d kpad=%d permuted=%d", nimg, H, W, kpad, row_src != nullptr);
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// E4M3 row quantisation for the fp8 wide GEMM (gemm256.hip): one wave per row, two passes over the row (the second
// one is an L2 hit): scale = max|x| / 448, Q = e4m3(x * (448 / max|x|)), round-to-nearest-even (v_cvt_pk_fp8_f32).
// ------------------------------------------------------------------------------------------------
struct QuantArgs {
  const bf16* x; unsigned char* q; float* scale;
  int rows, K, ldx, ldq;
};
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(QuantArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= a.rows) return;
  const int nch = a.K >> 3;
  const bf16* src = a.x + (long)row * a.ldx;
  float amax = 0.f;
  for (int ch = lane; ch < nch; ch += 64) {
    const bf16x8 v = *(const bf16x8*)(src + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(v[e])));
  }
  amax = wave_max(amax);
  const float inv = amax > 0.f ? 448.0f / amax : 0.f;
  if (lane == 0) a.scale[row] = amax > 0.f ? amax / 448.0f : 1.0f;
  unsigned char* dst = a.q + (long)row * a.ldq;
  for (int ch = lane; ch < nch; ch += 64) {
    *(int2*)(dst + ch * 8) = e4m3_pack8(*(const bf16x8*)(src + ch * 8), inv);
  }
}

extern "C" int hwocr_quant_rows_fp8(const void* X, void* Q, float* scale, int rows, int K, int ldx, int ldq,
                                    hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!X || !Q || !scale || rows <= 0 || K <= 0 || K % 8 || ldx % 8 || ldq % 8 || ldx < K || ldq < K) return HWOCR_EINVAL;
  QuantArgs a{(const bf16*)X, (unsigned char*)Q, scale, rows, K, ldx, ldq};
  HWOCR_PLAN("quant_rows_fp8_kernel rows=%d K=%d", rows, K);
  hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The prefill's fill of an E4M3 KV cache (hwocr_kv.fp8, head_dim 256): bf16 K rows / V^T rows of one prefill call and layer -> codes in
// the decode attention's operand order (common.h kv8_k / kv8_v) + one scale per token and kv head (scale = max|x| / 448 over the
// token's 256 features, 1 for an all-zero row; code = e4m3(x * 448 / max|x|): hwocr_quant_rows_fp8's rule).  One 256-thread workgroup per
// (32-key block, kv head, read).
struct KvQuantArgs {
  const bf16* K; const bf16* VT; long k_seq, k_head, v_seq, v_head, v_row;
  unsigned char* K8; unsigned char* V8; float* ks; float* vs;
  int Hkv, keys, ctx;
};
__global__ __launch_bounds__(256) void kv_quant_fp8_kernel(KvQuantArgs a) {
  constexpr int HD = 256;
  const int kb = blockIdx.x, hk = blockIdx.y, j = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  __shared__ float s_vmax[32];
  __shared__ float s_vinv[32];
  const long reg = ((long)j * a.Hkv + hk) * (long)a.ctx;
  unsigned char* K8 = a.K8 + reg * HD;
  unsigned char* V8 = a.V8 + reg * HD;
  if (tid < 32) s_vmax[tid] = 0.f;
  __syncthreads();
  // ---- keys: wave w takes keys 8 w .. 8 w + 7 of the block, a lane 4 consecutive features of a key
  const bf16* Kp = a.K + j * a.k_seq + hk * a.k_head;
  for (int i = 0; i < 8; ++i) {
    const int key = kb * 32 + 8 * w + i;
    const bf16x4 v = *(const bf16x4*)(Kp + (long)key * HD + 4 * lane);
    float amax = fmaxf(fmaxf(fabsf(bf2f(v[0])), fabsf(bf2f(v[1]))), fmaxf(fabsf(bf2f(v[2])), fabsf(bf2f(v[3]))));
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.0f / amax : 0.f;
    if (lane == 0) a.ks[reg + key] = amax > 0.f ? amax / 448.0f : 1.0f;
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(bf2f(v[e]) * inv, -448.0f), 448.0f);
    int code = 0;
    code = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], code, false);
    code = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], code, true);
    *(int*)(K8 + kv8_k(key, 4 * lane)) = code;   // features 4 lane .. 4 lane + 3 are 4 consecutive bytes of one 8-byte group
  }
  // ---- values: thread d holds row d of V^T, the block's 32 keys; per-key maxima through LDS
  const bf16* Vp = a.VT + j * a.v_seq + hk * a.v_head + (long)tid * a.v_row + kb * 32;
  bf16x8 vv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) vv[g] = *(const bf16x8*)(Vp + 8 * g);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e)   // non-negative floats order like their bit patterns
      atomicMax((int*)&s_vmax[8 * g + e], __float_as_int(fabsf(bf2f(vv[g][e]))));
  __syncthreads();
  if (tid < 32) {
    const float amax = s_vmax[tid];
    s_vinv[tid] = amax > 0.f ? 448.0f / amax : 0.f;
    a.vs[reg + kb * 32 + tid] = amax > 0.f ? amax / 448.0f : 1.0f;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) {   // keys 8 g .. 8 g + 7 of row tid: the 8 bytes at kv8_v(tid, 32 kb + 8 g)
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(bf2f(vv[g][e]) * s_vinv[8 * g + e], -448.0f), 448.0f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *(int2*)(V8 + kv8_v(tid, kb * 32 + 8 * g)) = make_int2(lo, hi);
  }
}

extern "C" int hwocr_kv_quant_fp8(const void* K, const void* VT, long k_seq, long k_head, long v_seq, long v_head, long v_row, void* K8,
                                  void* VT8, float* k_scale, float* v_scale, int nseq, int Hkv, int keys, int ctx, hipStream_t stream) {
  (void)hipGetLastError();
  if (!K || !VT || !K8 || !VT8 || !k_scale || !v_scale || nseq <= 0 || Hkv <= 0 || keys <= 0 || (keys % 32) || keys > ctx || (ctx % 32) ||
      (v_row % 8) || v_row < keys || (k_seq % 4) || (k_head % 4) || (v_seq % 8) || (v_head % 8))
    return HWOCR_EINVAL;
  KvQuantArgs a{(const bf16*)K, (const bf16*)VT, k_seq, k_head, v_seq, v_head, v_row, (unsigned char*)K8, (unsigned char*)VT8, k_scale,
                v_scale, Hkv, keys, ctx};
  HWOCR_PLAN("kv_quant_fp8_kernel nseq=%d Hkv=%d keys=%d ctx=%d", nseq, Hkv, keys, ctx);
  hipLaunchKernelGGL(kv_quant_fp8_kernel, dim3(keys / 32, Hkv, nseq), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

// the instance of a one-wave-per-row kernel whose chunk count fits the row: 1, 2, 3, 4 or 8 chunks of 8 elements per lane
struct LaunchLayerNorm {
  template <int NC> static void go(int grid, hipStream_t st, const LayerNormArgs& a) {
    hipLaunchKernelGGL(layernorm_kernel<NC>, dim3(grid), dim3(256), 0, st, a);
  }
};
struct LaunchAddRmsNorm {
  template <int NC> static void go(int grid, hipStream_t st, const RmsArgs& a) {
    hipLaunchKernelGGL(add_rmsnorm_kernel<NC>, dim3(grid), dim3(256), 0, st, a);
  }
};
static int chunks_instance(int D) {  // NC of the instance launch_by_chunks picks
  const int nc = (D / 8 + 63) / 64;
  return nc <= 1 ? 1 : nc <= 4 ? nc : MAXC;
}
template <typename L, typename Args>
static void launch_by_chunks(int D, int grid, hipStream_t st, const Args& a) {
  const int nc = (D / 8 + 63) / 64;
  if (nc <= 1) L::template go<1>(grid, st, a);
  else if (nc == 2) L::template go<2>(grid, st, a);
  else if (nc == 3) L::template go<3>(grid, st, a);
  else if (nc == 4) L::template go<4>(grid, st, a);
  else L::template go<MAXC>(grid, st, a);
}

static int ln_grid(int rows) {
  static const int cap = HWOCR_DIAG_ENV_INT("HWOCR_LN_GRID", 4096);  // (62208 x 1280, cold: 75 us uncapped, 70 at 4096, 83 at 2048)
  const int need = (rows + 3) / 4;
  return need < cap ? need : cap;
}

extern "C" int hwocr_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int D, int ldx,
                               int ldo, float eps, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (rows <= 0 || D % 8 || D > 64 * 8 * MAXC || ldx % 8 || ldo % 8) return HWOCR_EINVAL;
  LayerNormArgs a{(const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)out, rows, D, ldx, ldo, eps, nullptr, nullptr, 0};
  HWOCR_PLAN("layernorm_kernel<NC=%d> fp8=0 rows=%d D=%d grid=%d trips=%d", chunks_instance(D), rows, D, ln_grid(rows),
             ((rows + 3) / 4 + ln_grid(rows) - 1) / ln_grid(rows));
  launch_by_chunks<LaunchLayerNorm>(D, ln_grid(rows), stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_layernorm_fp8(const void* x, const void* w, const void* b, void* q8, float* q8s, int rows, int D,
                                   int ldx, int ldq, float eps, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!q8 || !q8s || rows <= 0 || D % 8 || D > 64 * 8 * MAXC || ldx % 8 || ldq % 8 || ldq < D) return HWOCR_EINVAL;
  LayerNormArgs a{(const bf16*)x, (const bf16*)w, (const bf16*)b, nullptr, rows, D, ldx, 0, eps, (unsigned char*)q8, q8s, ldq};
  HWOCR_PLAN("layernorm_kernel<NC=%d> fp8=1 rows=%d D=%d grid=%d trips=%d", chunks_instance(D), rows, D, ln_grid(rows),
             ((rows + 3) / 4 + ln_grid(rows) - 1) / ln_grid(rows));
  launch_by_chunks<LaunchLayerNorm>(D, ln_grid(rows), stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_rmsnorm_fp8(const void* h, int ldh, const void* w, void* q8, float* q8s, int ldq, int rows, int D,
                                 float eps, int gemma, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!q8 || !q8s || rows <= 0 || D % 8 || D > 64 * 8 * MAXC || ldh % 8 || ldq % 8 || ldq < D) return HWOCR_EINVAL;
  RmsArgs a{nullptr, 0, 0, 0, nullptr, (bf16*)h, ldh, (const bf16*)w, nullptr, 0, nullptr, rows, D, eps, gemma};
  a.q8 = (unsigned char*)q8; a.q8s = q8s; a.ldq = ldq;
  HWOCR_PLAN("add_rmsnorm_kernel<NC=%d> fp8=1 gemma=%d rows=%d D=%d nslab=0 gather=0", chunks_instance(D), gemma, rows, D);
  launch_by_chunks<LaunchAddRmsNorm>(D, (rows + 3) / 4, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_add_rmsnorm(const float* slabs, int nslab, long slab_stride, int ld_slab, const void* bias,
                                 void* h, int ldh, const void* w, void* out, int ldo, const int* row_index,
                                 int rows, int D, float eps, int gemma, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (rows <= 0 || D % 8 || D > 64 * 8 * MAXC || ldh % 8 || ldo % 8 || (nslab > 0 && (!slabs || ld_slab % 4)))
    return HWOCR_EINVAL;
  if (nslab > 0 && row_index) return HWOCR_EINVAL;
  RmsArgs a{slabs, nslab, slab_stride, ld_slab, (const bf16*)bias, (bf16*)h, ldh, (const bf16*)w, (bf16*)out, ldo,
            row_index, rows, D, eps, gemma};
  if (hwocr_plan_on()) {
    if (rows <= 512) hwocr_plan_note("add_rmsnorm_row_kernel<%d> gemma=%d rows=%d D=%d nslab=%d gather=%d", D <= 2048 ? 256 : 512, gemma, rows, D, nslab, row_index != nullptr);
    else hwocr_plan_note("add_rmsnorm_kernel<NC=%d> fp8=0 gemma=%d rows=%d D=%d nslab=%d gather=%d", chunks_instance(D), gemma, rows, D, nslab, row_index != nullptr);
    return HWOCR_OK;
  }
  if (rows <= 512 && D <= 2048)
    hipLaunchKernelGGL(add_rmsnorm_row_kernel<256>, dim3(rows), dim3(256), 0, stream, a);
  else if (rows <= 512)
    hipLaunchKernelGGL(add_rmsnorm_row_kernel<512>, dim3(rows), dim3(512), 0, stream, a);
  else
    launch_by_chunks<LaunchAddRmsNorm>(D, (rows + 3) / 4, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_vit_rope_split(const void* qkv, void* Q, void* K, void* VT, const int* pos_h, const int* pos_w,
                                    const float* cos_tab, const float* sin_tab, int tokens, int tok_ld, int heads,
                                    int hd, int interleaved, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (tokens <= 0 || tok_ld % 64 || tok_ld < tokens || hd % 16 || hd > 128) return HWOCR_EINVAL;
  VitRopeArgs a{(const bf16*)qkv, (bf16*)Q, (bf16*)K, (bf16*)VT, pos_h, pos_w, cos_tab, sin_tab,
                tokens, tok_ld, heads, hd, (long)tok_ld * hd, (long)hd * tok_ld, interleaved ? 1 : 0};
  HWOCR_PLAN("vit_rope_split_kernel tokens=%d heads=%d hd=%d interleaved=%d", tokens, heads, hd, interleaved ? 1 : 0);
  hipLaunchKernelGGL(vit_rope_split_kernel, dim3((tok_ld + 63) / 64, heads), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_mrope_kv_prefill(const void* qkv, void* Q, void* K, void* VT, const int* pos,
                                      const void* cos_tab, const void* sin_tab, int rows, int rows_per_seq, int Hq,
                                      int Hkv, int sec0, int sec1, long k_seq, long k_head, long v_seq, long v_head,
                                      long v_row, int head_dim, int kv_tiled, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (rows <= 0 || Hq <= 0 || Hkv <= 0 || rows_per_seq <= 0 || rows_per_seq > v_row) return HWOCR_EINVAL;
  if ((head_dim != 128 && head_dim != 256) || (kv_tiled && head_dim != 128)) return HWOCR_EINVAL;
  MropeArgs a{(const bf16*)qkv, (bf16*)Q, (bf16*)K, (bf16*)VT, pos, (const bf16*)cos_tab,
              (const bf16*)sin_tab, rows, rows_per_seq, Hq, Hkv, sec0, sec1, k_seq, k_head, v_seq, v_head, v_row,
              kv_tiled};
  HWOCR_PLAN("mrope_kv_prefill_kernel<%d> rows=%d Hq=%d Hkv=%d tiled=%d", head_dim, rows, Hq, Hkv, kv_tiled);
  if (head_dim == 128)
    hipLaunchKernelGGL(mrope_kv_prefill_kernel<128>, dim3((rows + 63) / 64, Hq + 2 * Hkv), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(mrope_kv_prefill_kernel<256>, dim3((rows + 63) / 64, Hq + 2 * Hkv), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_decode_qkv_finish(const float* slabs, int nslab, long slab_stride, const void* bias, void* Q,
                                       void* K, void* VT, const int* lens, const int* rope_delta,
                                       const void* cos_tab, const void* sin_tab, int nseq, int Hq, int Hkv,
                                       long k_seq, long k_head, long v_seq, long v_head, long v_row, int head_dim,
                                       int kv_tiled, int ctx, int max_pos, int* status, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (nseq <= 0 || nslab < 1 || !slabs || ctx < 1 || max_pos < 1) return HWOCR_EINVAL;
  if ((head_dim != 128 && head_dim != 256) || (kv_tiled && head_dim != 128)) return HWOCR_EINVAL;
  DecQkvArgs a{slabs, nslab, slab_stride, (const bf16*)bias, (bf16*)Q, (bf16*)K, (bf16*)VT, lens, rope_delta,
               (const bf16*)cos_tab, (const bf16*)sin_tab, Hq, Hkv, k_seq, k_head, v_seq, v_head, v_row, kv_tiled,
               ctx, max_pos, status};
  const size_t lds = (size_t)(Hq + 2 * Hkv) * head_dim * 2;
  HWOCR_PLAN("decode_qkv_finish_kernel<%d> nseq=%d Hq=%d Hkv=%d tiled=%d nslab=%d", head_dim, nseq, Hq, Hkv, kv_tiled, nslab);
  if (head_dim == 128) hipLaunchKernelGGL(decode_qkv_finish_kernel<128>, dim3(nseq), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(decode_qkv_finish_kernel<256>, dim3(nseq), dim3(256), lds, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_embed_splice(const int* ids, const int* img_row, const void* table, const void* img, void* out,
                                  int rows, int D, float scale, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (rows <= 0 || D % 8) return HWOCR_EINVAL;
  EmbedArgs a{ids, img_row, (const bf16*)table, (const bf16*)img, (bf16*)out, rows, D, scale};
  const long total = (long)rows * (D / 8);
  HWOCR_PLAN("embed_splice_kernel rows=%d D=%d spliced=%d", rows, D, img_row != nullptr);
  hipLaunchKernelGGL(embed_splice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
  return hwocr_launch_status();
}

extern "C" int hwocr_argmax_advance(const void* logits, int ldl, int V, int nseq, int* cur_ids, int* lens, int* n_gen,
                                    int* finished, int* out_tokens, int max_new, int min_new, const int* eos,
                                    int n_eos, int pad_id, unsigned* seen, int seen_ld, float rep_penalty, int* split_ws,
                                    hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (nseq <= 0 || V % 8 || ldl % 8 || n_eos < 0 || n_eos > 4) return HWOCR_EINVAL;
  if (seen && (seen_ld * 32 < V || !(rep_penalty > 0.f))) return HWOCR_EINVAL;
  SelectArgs a{(const bf16*)logits, ldl, V, cur_ids, lens, n_gen, finished, out_tokens, max_new, min_new,
               {0, 0, 0, 0}, n_eos, pad_id, (seen && rep_penalty != 1.0f) ? seen : nullptr, seen_ld, rep_penalty};
  static_assert(SEL_WS_INTS == HWOCR_SELECT_WS_INTS && 1 + 2 * SEL_PARTS <= SEL_WS_INTS, "hwocr.h: split_ws ints per read");
  const bool split = split_ws && nseq <= 16 && V >= 8 * 256 * SEL_PARTS;  // few reads, a row worth cutting
  HWOCR_PLAN("argmax_advance_kernel<%s> nseq=%d V=%d penalty=%d", split ? "256,16" : "1024,1", nseq, V, a.seen != nullptr);
  for (int k = 0; k < n_eos; ++k) a.eos[k] = eos[k];
  if (split) {
    a.split_ws = split_ws;
    hipLaunchKernelGGL((argmax_advance_kernel<256, SEL_PARTS>), dim3(nseq, SEL_PARTS), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((argmax_advance_kernel<ARG_THREADS, 1>), dim3(nseq), dim3(ARG_THREADS), 0, stream, a);
  }
  return hwocr_launch_status();
}

extern "C" int hwocr_sample_advance(const void* logits, int ldl, int V, int nseq, int* cur_ids, int* lens, int* n_gen,
                                    int* finished, int* out_tokens, int max_new, int min_new, const int* eos, int n_eos,
                                    int pad_id, unsigned* seen, int seen_ld, float rep_penalty, float temperature, int top_k,
                                    float top_p, unsigned long long seed, const int* read_ids, unsigned long long* debug,
                                    hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (nseq <= 0 || V % 8 || ldl % 8 || n_eos < 0 || n_eos > 4) return HWOCR_EINVAL;
  if (seen && (seen_ld * 32 < V || !(rep_penalty > 0.f))) return HWOCR_EINVAL;
  if (!(temperature > 0.f) || !(top_p > 0.f)) return HWOCR_EINVAL;
  SampleArgs a{{(const bf16*)logits, ldl, V, cur_ids, lens, n_gen, finished, out_tokens, max_new, min_new,
                {0, 0, 0, 0}, n_eos, pad_id, (seen && rep_penalty != 1.0f) ? seen : nullptr, seen_ld, rep_penalty},
               1.4426950408889634f / temperature, top_k, top_p, (unsigned)seed, (unsigned)(seed >> 32), read_ids, debug};
  HWOCR_PLAN("sample_advance_kernel nseq=%d V=%d top_k=%d top_p=%g", nseq, V, top_k, (double)top_p);
  for (int k = 0; k < n_eos; ++k) a.g.eos[k] = eos[k];
  hipLaunchKernelGGL(sample_advance_kernel, dim3(nseq), dim3(ARG_THREADS), 0, stream, a);
  return hwocr_launch_status();
}
