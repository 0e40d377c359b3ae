// Decode-step GEMMs with at most 16 reads in flight (one page = 3 reads: BASELINE config 2 as literally stated; the tail of a
// continuous batch).  out[rows <= 16][N] = x[rows][K] . W[N][K]^T, W in the fragment-tiled layout of hwocr_tile_weights.
//
// Why a kernel family of its own.  With a handful of rows a decode step is ~200 launches of 5-12 us each and nothing in them is
// bandwidth: the ring kernel of gemm_stream.hip walks K in 64-wide tiles with a 16-wave barrier per tile (24 trips for K = 1536),
// every projection needs a second launch to combine its split-K slabs and normalise (add_rmsnorm_row, 4.9 us x 57 per token), and
// the r03c profile of a 3-read step shows 2.07 ms per token for 3.09 GB = 1.5 TB/s.  Here:
//   * no ring and no per-K-tile barrier: a workgroup of 16 waves owns a few weight tiles; the 16 / TC waves that share a tile split
//     its K range and each streams its slice HBM -> VGPR with every load of a chunk in flight (non-temporal: a weight byte is read
//     once per step), multiplies (one MFMA 16x16x32 per 1-KiB fragment, all <= 16 rows at once) and the slices are summed
//     through LDS in ascending K order (deterministic);
//   * the RMSNorm in front of the QKV and gate/up projections is the kernel's PROLOGUE: wave w normalises row w (the reference's
//     rounding chain, add_rmsnorm_kernel's arithmetic) into an LDS image of x that the B fragments are read from — every workgroup
//     recomputes it (<= 16 rows x 3-7 KB from L2), which is cheaper than a launch; the residual update that add_rmsnorm did on the
//     way (h <- bf16(bf16(sum of the down projection's slabs) + h)) is done there too and written back ONCE (workgroup (0, 0)) to
//     the OTHER residual buffer (ping-pong: every workgroup reads the old one);
//   * the output projection adds the residual in its epilogue (in place: an element is read and written by one lane), so it needs
//     no slabs; gate/up applies SwiGLU / GeGLU in its epilogue as everywhere else; only the long-K down projection still splits K
//     over workgroups (to reach every CU) and leaves fp32 slabs, which the NEXT layer's QKV prologue (or the final norm) sums.
// A decoder layer is 5 launches instead of 8 (qkv, attention split with the last workgroup merging, o, gate/up, down; 6 when the
// merge is a launch of its own, HWOCR_DECODE_LASTWG=0); hwocr_decode_step takes this path
// at <= 16 reads when every weight has its bf16 fragment-tiled copy.
#include "gemm_common.h"

using namespace gemm;

namespace {

constexpr int R16_WAVES = 16;
// k-steps (1-KiB weight fragments) a wave keeps in flight / in registers at a time: 12, or 8 in the instance whose norm prologue holds
// rows of up to 4096 elements (a 16-wave workgroup leaves a lane 128 registers)
constexpr int rows16_chunk(int nc) { return nc > 4 ? 8 : 12; }
constexpr int R16_RED = R16_WAVES * 1024;

struct Rows16Args {
  const bf16* X; int ldx;           // NORM == false: activation rows [Bsz][K]
  const bf16* W;                    // [N/16][K/32][64 lanes][8]
  void* out; int ldo;               // PARTIAL: fp32 slabs [grid.y][Bsz][ldo]; RESIDUAL: bf16 rows updated in place; GLU: bf16 [Bsz][ldo]
  int Bsz, N, K;
  int ksteps_per_slice;             // K range of a workgroup in 32-wide k-steps (split-K over grid.y; RESIDUAL / GLU: the whole K)
  int tc;                           // tiles a workgroup multiplies at once (1, 2, 4, 8 or 16): 16 / tc waves share a tile's K range
  // NORM: x = RMSNorm(h'), h' = bf16(bf16(sum of slabs) + h_in) (nslab == 0: h' = h_in); workgroup (0, 0) stores h' to h_out
  const bf16* h_in; bf16* h_out; int ldh;
  const float* slabs; int nslab; long slab_stride; int ld_slab;
  const bf16* norm_w; float eps; int gemma;
};

// NC: 64-lane chunks of 8 elements a row of the norm prologue may have (4: K <= 2048, 8: K <= 4096); 0: no prologue
template <int EPI, int NC>
__global__ __launch_bounds__(64 * R16_WAVES) void gemm_rows16_kernel(Rows16Args a) {
  constexpr bool NORM = NC > 0;
  constexpr int R16_CH = rows16_chunk(NC);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;                 // [16 waves][64 lanes][4]: K-slice partial sums
  char* xs = smem + R16_RED;                 // NORM: x image, row r at r * xstride (16 bytes of padding per row: the B-fragment
  const int xstride = 2 * a.K + 16;          // reads of a 16-lane group then touch different banks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;

  constexpr int UNIT = is_glu<EPI> ? 2 : 1;
  const int units = (a.N >> 4) / UNIT;
  const int t0 = (int)((long)blockIdx.x * units / gridDim.x) * UNIT;
  const int t1 = (int)((long)(blockIdx.x + 1) * units / gridDim.x) * UNIT;
  const int TC = a.tc, KS = R16_WAVES / TC;   // powers of two
  const int wt = w / KS, wk = w - wt * KS;    // tile slot of this wave, its K slice
  const int ks_total = a.K >> 5;
  const int ks0 = blockIdx.y * a.ksteps_per_slice;
  const int nks = min(a.ksteps_per_slice, ks_total - ks0);
  const int kb = ks0 + (int)((long)wk * nks / KS), ke = ks0 + (int)((long)(wk + 1) * nks / KS);
  const bool through_lds = KS > 1 || is_glu<EPI>;

  // The first chunk of this wave's weight fragments goes out BEFORE the norm prologue: the loads do not depend on x, and their HBM
  // latency (the longest thing in the kernel at these sizes) then passes under the prologue instead of after it.
  bf16x8 wf[R16_CH];
  const bool on0 = t0 + wt < t1 && ke > kb;   // wave-uniform
  // (loads go out in groups of four k-steps: a group is skipped - a wave-uniform branch - when the slice ends before it, so a short
  // slice does not re-request its last fragment up to eleven times; inside a group indices are clamped, never a branch per load)
  auto load_w = [&](const bf16* wp, int k) {
#pragma unroll
    for (int g = 0; g < R16_CH; g += 4)
      if (k + g < ke) {
#pragma unroll
        for (int i = g; i < g + 4; ++i) wf[i] = __builtin_nontemporal_load((const bf16x8*)(wp + (size_t)min(k + i, ke - 1) * 512));
      }
  };
  if (on0) load_w(a.W + (size_t)(t0 + wt) * ks_total * 512 + lane * 8, kb);
  // the residual rows of the first round's epilogue likewise: nothing they depend on is computed here
  bf16x4 res0 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  if constexpr (EPI == EPI_RESIDUAL) {
    if (wk == 0 && t0 + wt < t1 && c < a.Bsz) res0 = *(const bf16x4*)((const bf16*)a.out + (size_t)c * a.ldo + 16 * (t0 + wt) + 4 * q);
  }

  if constexpr (NORM) {
    const int D = a.K, nch = D >> 3;
    if (w < a.Bsz) {
      // NC chunks of 8 per lane, kept as the bf16 values they are (4 registers a chunk: the weight fragments already in flight take
      // 32-48 of the 128 a 16-wave workgroup leaves a lane)
      bf16x8 hs[NORM ? NC : 1];
      float ss = 0.f;
      const bool keeper = blockIdx.x == 0 && blockIdx.y == 0 && a.h_out != nullptr;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
          bf16x8 hv = *(const bf16x8*)(a.h_in + (long)w * a.ldh + ch * 8);
          if (a.nslab > 0) {
            // the (<= 4) slabs two at a time, loads unconditional (a slab past nslab re-reads slab 0 and adds zero), summed in
            // ascending slab order
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = 0.f;
#pragma unroll
            for (int s0 = 0; s0 < 4; s0 += 2) {
              f32x4 v[2][2];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const float* p = a.slabs + (long)(s0 + u < a.nslab ? s0 + u : 0) * a.slab_stride + (long)w * a.ld_slab + ch * 8;
                v[u][0] = *(const f32x4*)p;
                v[u][1] = *(const f32x4*)(p + 4);
              }
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const float on = s0 + u < a.nslab ? 1.0f : 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[e] += on * v[u][0][e]; y[4 + e] += on * v[u][1][e]; }
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = f2bf(rbf(y[e]) + bf2f(hv[e]));
          }
          if (keeper) *(bf16x8*)(a.h_out + (long)w * a.ldh + ch * 8) = hv;
          hs[i] = hv;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float xv = bf2f(hv[e]); ss += xv * xv; }
        }
      }
      bf16x8 gw[NORM ? NC : 1];   // the norm weights, requested before the row statistic is reduced
#pragma unroll
      for (int i = 0; i < NC; ++i)
        if (lane + 64 * i < nch) gw[i] = *(const bf16x8*)(a.norm_w + (lane + 64 * i) * 8);
      const float rstd = rsqrtf(wave_sum(ss) / D + a.eps);
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
          const bf16x8 g = gw[i];
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xv = bf2f(hs[i][e]);
            o[e] = a.gemma ? f2bf(xv * rstd * (1.0f + bf2f(g[e]))) : f2bf(bf2f(g[e]) * rbf(xv * rstd));
          }
          *(bf16x8*)(xs + w * xstride + ch * 16) = o;
        }
      }
    } else {  // rows past Bsz: zeros (their output columns are never stored)
      const bf16x8 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      for (int ch = lane; ch < nch; ch += 64) *(bf16x8*)(xs + w * xstride + ch * 16) = z;
    }
    __syncthreads();
  }


  for (int tbase = t0; tbase < t1; tbase += TC) {  // workgroup-uniform
    const int tile = tbase + wt;
    const bool mine = tile < t1;                    // wave-uniform: this wave has a tile in this round ...
    const bool on = mine && ke > kb;                // ... and its K slice is not empty (fewer k-steps than slices: tiny models)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (on) {
      const bf16* wp = a.W + (size_t)tile * ks_total * 512 + lane * 8;
      const bf16* xp = NORM ? nullptr : a.X + (size_t)min(c, a.Bsz - 1) * a.ldx + q * 8;
      for (int k = kb; k < ke; k += R16_CH) {
        if (tbase != t0 || k != kb) load_w(wp, k);  // (the first chunk of the first round is already in flight)
        if constexpr (NORM) {
#pragma unroll
          for (int i0 = 0; i0 < R16_CH; i0 += 4) {  // x fragments from the LDS image, four at a time
            if (k + i0 >= ke) break;
            bf16x8 xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(xs + c * xstride + min(k + i0 + i, ke - 1) * 64 + q * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k + i0 + i < ke) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i0 + i], xf[i], acc, 0, 0, 0);
          }
        } else {
          bf16x8 xf[R16_CH];
#pragma unroll
          for (int g = 0; g < R16_CH; g += 4)
            if (k + g < ke) {
#pragma unroll
              for (int i = g; i < g + 4; ++i) xf[i] = *(const bf16x8*)(xp + min(k + i, ke - 1) * 32);
            }
#pragma unroll
          for (int i = 0; i < R16_CH; ++i)
            if (k + i < ke) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[i], acc, 0, 0, 0);
        }
      }
    }
    // ---- the K slices of a tile summed in ascending order by the wave that holds slice 0; a gated pair by the gate tile's wave
    f32x4 up = f32x4{0.f, 0.f, 0.f, 0.f};
    if (through_lds) {
      *(f32x4*)(red + (w * 64 + lane) * 4) = acc;
      __syncthreads();
      if (wk == 0 && mine) {
        for (int j = 1; j < KS; ++j) acc += *(const f32x4*)(red + ((w + j) * 64 + lane) * 4);
        if constexpr (is_glu<EPI>) {
          if ((wt & 1) == 0)
            for (int j = 0; j < KS; ++j) up += *(const f32x4*)(red + ((w + KS + j) * 64 + lane) * 4);
        }
      }
    }
    // lane (c, q): acc[r] = out[row c][16 tile + 4 q + r]
    if (wk == 0 && mine && c < a.Bsz) {
      const int n = 16 * tile + 4 * q;
      if constexpr (EPI == EPI_PARTIAL) {
        *(f32x4*)((float*)a.out + ((size_t)blockIdx.y * a.Bsz + c) * a.ldo + n) = acc;
      } else if constexpr (EPI == EPI_RESIDUAL) {
        bf16* hp = (bf16*)a.out + (size_t)c * a.ldo + n;
        const bf16x4 h = tbase == t0 ? res0 : *(const bf16x4*)hp;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(acc[r]) + bf2f(h[r]));
        *(bf16x4*)hp = o;
      } else if constexpr (is_glu<EPI>) {
        if ((wt & 1) == 0) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(glu_gate<EPI>(rbf(acc[r]))) * rbf(up[r]));
          *(bf16x4*)((bf16*)a.out + (size_t)c * a.ldo + 8 * tile + 4 * q) = o;
        }
      } else {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[r]);
        *(bf16x4*)((bf16*)a.out + (size_t)c * a.ldo + n) = o;
      }
    }
    if (through_lds) __syncthreads();  // the next round overwrites the partial sums
  }
}

template <int EPI, int NC>
int launch_rows16_nc(const Rows16Args& a, dim3 grid, hipStream_t st) {
  const int lds = R16_RED + (NC ? 16 * (2 * a.K + 16) : 0);
  static const bool attr_done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_rows16_kernel<EPI, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_done;
  hipLaunchKernelGGL((gemm_rows16_kernel<EPI, NC>), grid, dim3(64 * R16_WAVES), lds, st, a);
  return hwocr_launch_status();
}
template <int EPI, bool NORM>
int launch_rows16(const Rows16Args& a, dim3 grid, hipStream_t st) {
  if constexpr (!NORM) return launch_rows16_nc<EPI, 0>(a, grid, st);
  else return a.K <= 2048 ? launch_rows16_nc<EPI, 4>(a, grid, st) : launch_rows16_nc<EPI, 8>(a, grid, st);
}

int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

// norm: the hwocr_rows16_norm block or NULL.  splitk > 1 only with HWOCR_EPI_PARTIAL.  See include/hwocr.h.
extern "C" int hwocr_gemm_rows16(const void* X, int ldx, const void* Wt, void* out, int ldo, int Bsz, int N, int K, int epi, int splitk,
                                 const hwocr_rows16_norm* norm, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!Wt || !out || Bsz < 1 || Bsz > 16 || N <= 0 || K <= 0 || (N % 16) || (K % 32) || (ldo % 4) || splitk < 1) return HWOCR_EINVAL;
  const bool glu = epi == EPI_SWIGLU || epi == EPI_GEGLU;
  if (epi != EPI_PARTIAL && epi != EPI_RESIDUAL && epi != EPI_LINEAR && !glu) return HWOCR_EINVAL;
  if (glu && (N % 32)) return HWOCR_EINVAL;
  if (epi != EPI_PARTIAL && splitk != 1) return HWOCR_EINVAL;
  if (norm) {
    if (!norm->h_in || !norm->norm_w || (norm->ldh % 8) || K > 4096 || (K % 8) || norm->nslab < 0 || norm->nslab > 4 ||
        (norm->nslab > 0 && (!norm->slabs || (norm->ld_slab % 4))))
      return HWOCR_EINVAL;
  } else if (!X || (ldx % 8)) {
    return HWOCR_EINVAL;
  }
  const int ks_total = K / 32;
  const int per = (ks_total + splitk - 1) / splitk;
  if ((splitk - 1) * per >= ks_total) return HWOCR_EINVAL;  // an empty slice would leave its slab unwritten
  const int unit = glu ? 2 : 1, units = N / 16 / unit;
  const int groups = units < 256 / splitk ? units : (256 / splitk > 0 ? 256 / splitk : 1);
  const int tiles_per_wg = ((units + groups - 1) / groups) * unit;
  int tc = pow2_at_least(tiles_per_wg);
  if (tc > 16) tc = 16;
  if (glu && tc < 2) tc = 2;
  Rows16Args a{(const bf16*)X, ldx, (const bf16*)Wt, out, ldo, Bsz, N, K, per, tc,
               norm ? (const bf16*)norm->h_in : nullptr, norm ? (bf16*)norm->h_out : nullptr, norm ? norm->ldh : 0,
               norm ? norm->slabs : nullptr, norm ? norm->nslab : 0, norm ? norm->slab_stride : 0, norm ? norm->ld_slab : 0,
               norm ? (const bf16*)norm->norm_w : nullptr, norm ? norm->eps : 0.f, norm ? norm->gemma : 0};
  const dim3 grid(groups, splitk);
  HWOCR_PLAN("gemm_rows16_kernel<epi=%d,%s> tc=%d %s rows=%d N=%d K=%d splitk=%d groups=%d nslab=%d", epi, norm ? (K <= 2048 ? "norm4" : "norm8") : "plain", tc,
             tiles_per_wg > tc ? "rounds>1" : "rounds=1", Bsz, N, K, splitk, groups, norm ? norm->nslab : 0);
  if (norm) {
    switch (epi) {
      case EPI_PARTIAL: return launch_rows16<EPI_PARTIAL, true>(a, grid, stream);
      case EPI_SWIGLU: return launch_rows16<EPI_SWIGLU, true>(a, grid, stream);
      case EPI_GEGLU: return launch_rows16<EPI_GEGLU, true>(a, grid, stream);
      case EPI_LINEAR: return launch_rows16<EPI_LINEAR, true>(a, grid, stream);
      default: return HWOCR_EINVAL;  // a normalised input with the residual epilogue does not occur in a decoder layer
    }
  }
  switch (epi) {
    case EPI_PARTIAL: return launch_rows16<EPI_PARTIAL, false>(a, grid, stream);
    case EPI_RESIDUAL: return launch_rows16<EPI_RESIDUAL, false>(a, grid, stream);
    case EPI_SWIGLU: return launch_rows16<EPI_SWIGLU, false>(a, grid, stream);
    case EPI_GEGLU: return launch_rows16<EPI_GEGLU, false>(a, grid, stream);
    case EPI_LINEAR: return launch_rows16<EPI_LINEAR, false>(a, grid, stream);
    default: return HWOCR_EINVAL;
  }
}
