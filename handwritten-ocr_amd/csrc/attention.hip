// Attention kernels of the page-read path (gfx950 / MI355X).
//
//  attn_prefill : flash-style attention over whole segments (one page image in the vision tower,
//                 one prompt in decoder prefill).  Replaces F.scaled_dot_product_attention as reached
//                 from HF modeling_qwen2_vl.py:375-418 (vision, non-causal, one segment per image) and
//                 :553-569 (decoder prefill, causal, GQA).  The score tile is computed TRANSPOSED
//                 (S^T = K.Q^T, 32x32x16 MFMA) so every lane owns one query column: softmax statistics
//                 are lane-local (+1 exchange with lane^32) and the fp32 score registers, packed to bf16,
//                 are directly the B operand of O^T += V^T.P^T — P never touches LDS.  V is consumed
//                 through a transposed image V^T[d][key] that the rope/split kernels write.
//  attn_decode  : one new token per read against its KV cache (HBM-bound).  K rows and V^T rows go
//                 straight from HBM to MFMA operands; keys are split over waves and workgroups and merged
//                 with the usual (m, l, O) rule.
#include "attention_args.h"
#include "common.h"
#include "hwocr.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

using namespace hwocr_attn;

namespace {

__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, HD > 128 ? 1 : 2) void attn_prefill_kernel(PrefillArgs a) {
  constexpr int HDP = (HD + 31) / 32 * 32;  // d rows of the V^T tile, padded to whole 32-row MFMA tiles
  constexpr int KS = HD * 2 + 16;           // K tile row stride (bytes): +1 slot -> conflict-free ds_read_b128
  constexpr int VS = 144;                   // V^T tile row stride (bytes): 64 keys + 16 B -> conflict-free ds_read_b128
  constexpr int KBYTES = 64 * KS, VBYTES = HDP * VS;
  constexpr int NS = HD / 16, ND = HDP / 32;
  constexpr int CK = 64 * HD / 8, CV = HD * 8;             // 16-byte chunks per tile
  constexpr int NKC = (CK + 255) / 256, NVC = (CV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int seg = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
  const int len = a.lens[seg];
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int hk = h / a.group;
  const long off = a.seg_off ? a.seg_off[seg] : 0;  // varlen: rows of all segments share one packed buffer
  const bf16* Qp = a.Q + seg * a.q_seg + h * a.q_head + off * a.q_row;
  const bf16* Kp = a.K + seg * a.k_seg + hk * a.k_head + off * a.k_row;
  const bf16* Vp = a.VT + seg * a.v_seg + hk * a.v_head + off;

  const int qi = q0 + 32 * w + r;  // this lane's query column
  const int rk = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // r with bits 2 and 3 swapped
  // query fragments: in registers, except for 256-wide heads (64 VGPRs that push the kernel into spills): those re-read
  // their 16-byte pieces from L1/L2 inside the score loop
  constexpr bool QREG = HD <= 128;
  const bf16* qrow = Qp + (long)min(qi, len - 1) * a.q_row + 8 * hh;
  bf16x8 qf[QREG ? NS : 1];
  if constexpr (QREG) {
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8*)(qrow + 16 * s);
  }

  const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
  const int nt = (kv_end + 63) >> 6;

  bf16x8 kreg[NKC], vreg[NVC];
  auto load_tile = [&](int t) {
    const int j0 = t * 64;
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + 256 * i;
      if (id < CK) {
        const int row = id / (HD / 8), ch = id % (HD / 8);
        const int key = min(j0 + row, len - 1);
        kreg[i] = *(const bf16x8*)(Kp + (a.kv_tiled ? kv_tiled_k(key, ch * 8) : (long)key * a.k_row + ch * 8));
      }
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + 256 * i;
      if (id < CV) {
        const int d = id >> 3, ch = id & 7;
        const bf16* vp = Vp + (a.kv_tiled ? kv_tiled_v(d, j0 + ch * 8) : (long)d * a.v_row + j0 + ch * 8);
        // a ragged segment starts on a 4-key (8-byte) boundary only
        bf16x8 v = a.seg_off ? cat4(*(const bf16x4*)vp, *(const bf16x4*)(vp + 4)) : *(const bf16x8*)vp;
        if (j0 + 64 > len) {  // tail tile: keys past the segment carry p = 0, keep 0 * x finite
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (j0 + ch * 8 + e >= len) v[e] = (bf16)0.0f;
        }
        vreg[i] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* kb = smem + buf * (KBYTES + VBYTES);
    char* vb = kb + KBYTES;
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + 256 * i;
      if (id < CK) {
        const int row = id / (HD / 8), ch = id % (HD / 8);
        *(bf16x8*)(kb + row * KS + ch * 16) = kreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + 256 * i;
      if (id < CV) {
        const int d = id >> 3, ch = id & 7;
        *(bf16x8*)(vb + d * VS + ch * 16) = vreg[i];
      }
    }
  };

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = NEG_BIG, l = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * 64;
    if (t + 1 < nt) load_tile(t + 1);
    const char* kb = smem + (t & 1) * (KBYTES + VBYTES);
    const char* vb = kb + KBYTES;

    // ---- S^T = K . Q^T : rows = keys (registers), column = this lane's query.  MFMA tile row rho is fed key
    // swap23(rho) (bits 2 and 3 exchanged), so register i of lane-half hh holds key 16(i>>3) + 8hh + (i&7):
    // 8 consecutive keys per k-step, i.e. the natural B-operand order for the PV product below.
    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf16x8 k0 = *(const bf16x8*)(kb + rk * KS + (16 * s + 8 * hh) * 2);
      const bf16x8 k1 = *(const bf16x8*)(kb + (32 + rk) * KS + (16 * s + 8 * hh) * 2);
      const bf16x8 qs = QREG ? qf[QREG ? s : 0] : *(const bf16x8*)(qrow + 16 * s);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qs, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qs, s1, 0, 0, 0);
    }
    const bool need_mask = (j0 + 64 > len) || (CAUSAL && (j0 + 63 > q0 + 32 * w));
    if (need_mask) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = j0 + 16 * (i >> 3) + 8 * hh + (i & 7);
        if (key >= len || (CAUSAL && key > qi)) s0[i] = -INFINITY;
        if (key + 32 >= len || (CAUSAL && key + 32 > qi)) s1[i] = -INFINITY;
      }
    }
    float mx = NEG_BIG;  // raw (unscaled) scores; the positive scale commutes with max
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;
    // running-max update only when some query of this wave saw a larger score (exact: alpha == 1 otherwise)
    if (__any(mx > m)) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
      m = m_new;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s0[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[i], a.scale_log2, -m));
      s1[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], a.scale_log2, -m));
      rs += s0[i] + s1[i];
    }
    l += rs;

    // ---- P^T as B operand: k-step (kt, s2) element j <-> key 32kt + 16s2 + 8hh + j
    bf16x8 pb[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[0][s2][j] = f2bf(s0[8 * s2 + j]);
        pb[1][s2][j] = f2bf(s1[8 * s2 + j]);
      }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 vf = *(const bf16x8*)(vb + (d * 32 + r) * VS + (kt * 32 + 16 * s2 + 8 * hh) * 2);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kt][s2], o[d], 0, 0, 0);
        }

    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }

  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  if (qi < len) {
    bf16* orow = a.O + seg * a.o_seg + (off + qi) * a.o_row + h * HD;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * hh;
        if (dd < HD) {
          bf16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][4 * g + e] * inv);
          *(bf16x4*)(orow + dd) = ov;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Gemma prefill specialisation: head_dim 256, non-causal (PaliGemma's bidirectional prefix), row-major K / V^T, MQA/GQA.
// Same mathematics as attn_prefill_kernel<256,false>; what changes is how operands reach the matrix pipe:
//   * K / V^T tiles staged by LDS-DMA (the generic kernel spends 128 VGPRs on global->register->LDS staging), so the
//     query fragments (64 VGPRs) live in registers instead of being re-read from L2 every k-step;
//   * WAVES x 32 queries per workgroup stream one K / V^T: 8 waves halve the bytes entering the CU per query;
//   * swizzled LDS images (permutation applied to the DMA SOURCE address):
//       K   : [64 keys][512 B], 16-byte chunk p of row r holds chunk p ^ (r & 15)
//       V^T : [256 d  ][128 B], chunk p of row d holds chunk p ^ ((d >> 1) & 7)
//   * 1-D grid dealt so that every XCD owns whole reads: all query blocks and all query heads that share one read's
//     K / V^T (8.4 MB at 4113 tokens) share one L2.
// ------------------------------------------------------------------------------------------------
constexpr int G256_K = 64 * 512, G256_VT = 256 * 128, G256_STAGE = G256_K + G256_VT;  // 64 KiB per stage

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void attn_hd256_kernel(PrefillArgs a) {
  constexpr int HD = 256, NT = 64 * WAVES, NS = 16, ND = 8;
  constexpr int N_DMA = 64 / WAVES;  // 1-KiB DMA instructions per wave per tile: 32 for K (2 key rows each), 32 for V^T (8 d rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages
  int seg, h, q0;
  {
    const int per_seg = a.heads * a.qblocks, L = blockIdx.x;  // qblocks: blocks of 32 * WAVES queries
    int rem;
    if ((a.nseg & 7) == 0) {
      const int slot = L >> 3;
      seg = (slot / per_seg) * 8 + (L & 7);
      rem = slot % per_seg;
    } else {
      seg = L / per_seg;
      rem = L % per_seg;
    }
    h = rem % a.heads;  // the heads of one query block run together: same K / V^T tiles, in step
    q0 = (rem / a.heads) * (32 * WAVES);
  }
  const int len = a.lens[seg];
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int hk = h / a.group;
  const bf16* Qp = a.Q + seg * a.q_seg + h * a.q_head;
  const bf16* Kp = a.K + seg * a.k_seg + hk * a.k_head;
  const bf16* Vp = a.VT + seg * a.v_seg + hk * a.v_head;
  const int qi = q0 + 32 * w + r;
  const int rk = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // tile row -> key permutation (bits 2,3 swapped)

  bf16x8 qf[NS];
  {
    const bf16* qrow = Qp + (long)min(qi, len - 1) * a.q_row + 8 * hh;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8*)(qrow + 16 * s);
  }

  const int nt = (len + 63) >> 6;
  // DMA plan per tile: 64 instructions of 1 KiB, task k = w + WAVES e.  k < 32: K rows 2k, 2k+1; else V^T rows 8(k-32)..+7.
  // The lane part of every source address is tile- and task-independent (two variants for K: the swizzle sees row & 15,
  // and rows of successive tasks of a wave differ by 2 WAVES), the rest is wave-uniform.  K rows up to the end of the
  // last 64-key tile are read unclamped (their scores are masked), like the V^T columns: the cache is 64-aligned.
  const int krow = 2 * w + (lane >> 5);
  const long kofs0 = (long)krow * a.k_row + (((lane & 31) ^ (krow & 15)) << 3);
  const long kofs1 = (long)krow * a.k_row + (((lane & 31) ^ ((krow + 2 * WAVES) & 15)) << 3);
  const int vd = 8 * w + (lane >> 3);
  const long vofs = (long)vd * a.v_row + (((lane & 7) ^ ((vd >> 1) & 7)) << 3);
  static_assert(WAVES == 4 || WAVES == 8, "task -> row arithmetic below assumes 2 WAVES | 16 and 8 WAVES | 64");
  // piece i of this wave's N_DMA DMA instructions for tile t: the first half K rows, the second half V^T rows
  auto stage_piece = [&](int t, int i) {
    const int j0 = t * 64;
    char* st = smem + (t & 1) * G256_STAGE;
    if (i < N_DMA / 2) {  // K: task w + WAVES e -> rows krow + 2 WAVES e
      const int e = i;
      const bf16* src = Kp + (long)(j0 + 2 * WAVES * e) * a.k_row + ((e & 1) && WAVES == 4 ? kofs1 : kofs0);
      __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR(st + (w + WAVES * e) * 1024), 16, 0, 0);
    } else {              // V^T: task 32 + w + WAVES e -> d rows vd + 8 WAVES e (same swizzle: 8 WAVES e / 2 = 0 mod 8)
      const int e = i - N_DMA / 2;
      const bf16* src = Vp + (long)(8 * WAVES * e) * a.v_row + j0 + vofs;
      __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR(st + G256_K + (w + WAVES * e) * 1024), 16, 0, 0);
    }
  };
  auto stage_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) stage_piece(t, i);
  };

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = NEG_BIG, l = 0.f;

  stage_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * 64;
    char* st = smem + (t & 1) * G256_STAGE;
    // Tile t + 1 goes into the stage tile t - 1 occupied (every wave passed the barrier that ended it) piece by piece, one DMA
    // instruction behind each pair of score MFMAs below: a lone wave per SIMD pays ~100 cycles to ISSUE an LDS-DMA instruction
    // (tools/bench_vit80x_stamps.py), and the 16 of a tile issued in one burst at the top of the step held the matrix pipe idle
    // for 1600 of the step's ~6000 cycles.
    const bool more = t + 1 < nt;
    if (j0 + 64 > len) {
      // tail tile: keys past the read must contribute 0 * finite; their V^T columns are whatever the cache holds
      for (int i = tid; i < HD * 64; i += NT) {
        const int d = i >> 6, col = i & 63;
        if (j0 + col >= len) {
          const int ch = (col >> 3) ^ ((d >> 1) & 7);
          ((bf16*)(st + G256_K + d * 128 + ch * 16))[col & 7] = (bf16)0.0f;
        }
      }
      __syncthreads();
    }
    const char* kb = st;
    const char* vb = st + G256_K;

    // One wave per SIMD: nobody else hides the LDS latency, so a rolling window of fragments is kept 4 k-steps (8 MFMAs)
    // ahead of the MFMA that consumes them; every slot is refilled right behind its use (sched_barrier pins the order).
    auto k_frag = [&](int s, int half) {
      const int row = 32 * half + rk;
      return *(const bf16x8*)(kb + row * 512 + (((2 * s + hh) ^ (row & 15)) << 4));
    };
    auto v_frag = [&](int i) {  // i = 8 (2 kt + s2) + d: consecutive MFMAs of the PV product go to DIFFERENT accumulators
      const int dr = (i & 7) * 32 + r;
      return *(const bf16x8*)(vb + dr * 128 + ((((i >> 3) * 2 + hh) ^ ((dr >> 1) & 7)) << 4));
    };
    bf16x8 fr[8];
    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) fr[e] = k_frag(e >> 1, e & 1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * (s & 3)], qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * (s & 3) + 1], qf[s], s1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more && s < N_DMA) stage_piece(t + 1, s);
      if (s + 4 < NS) {
        fr[2 * (s & 3)] = k_frag(s + 4, 0);
        fr[2 * (s & 3) + 1] = k_frag(s + 4, 1);
      } else {  // the first V^T fragments land under the softmax
        fr[2 * (s & 3)] = v_frag(2 * (s & 3));
        fr[2 * (s & 3) + 1] = v_frag(2 * (s & 3) + 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (j0 + 64 > len) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = j0 + 16 * (i >> 3) + 8 * hh + (i & 7);
        if (key >= len) s0[i] = -INFINITY;
        if (key + 32 >= len) s1[i] = -INFINITY;
      }
    }
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s0[i]), s1[i]);  // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;
    if (__any(mx > m + a.slack)) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
      m = m_new;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s0[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[i], a.scale_log2, -m));
      s1[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], a.scale_log2, -m));
      rs += s0[i] + s1[i];
    }
    l += rs;
    bf16x8 pb[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[0][s2][j] = f2bf(s0[8 * s2 + j]);
        pb[1][s2][j] = f2bf(s1[8 * s2 + j]);
      }
    // O^T += V^T . P^T: fragment i = 8 (2 kt + s2) + d, window of 8
#pragma unroll
    for (int i = 0; i < 4 * ND; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      o[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i & 7], pb[i >> 4][(i >> 3) & 1], o[i & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 8 < 4 * ND) fr[i & 7] = v_frag(i + 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has had the whole multiply to land
    __syncthreads();
  }

  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  if (qi < len) {
    bf16* orow = a.O + seg * a.o_seg + (long)qi * a.o_row + h * HD;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][4 * g + e] * inv);
        *(bf16x4*)(orow + d * 32 + 8 * g + 4 * hh) = ov;
      }
  }
}

// log2 slack of the lazy running-max update: 8 = weights up to 256 (0 = exact schedule; HWOCR_ATTN_SLACK in the diagnostic build)
inline float attn_slack() {
  static const float v = HWOCR_DIAG_ENV_FLOAT("HWOCR_ATTN_SLACK", 8.0f);
  return v;
}

template <int WAVES>
int launch_hd256(PrefillArgs a, int nseg, int heads, int max_len, hipStream_t st) {
  HWOCR_PLAN("attn_hd256_kernel<%d> nseg=%d heads=%d group=%d max_len=%d", WAVES, nseg, heads, a.group, max_len);
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    hipFuncSetAttribute((const void*)attn_hd256_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G256_STAGE);
    return true;
  }();
  (void)done;
  a.qblocks = (max_len + 32 * WAVES - 1) / (32 * WAVES);
  a.slack = attn_slack();
  hipLaunchKernelGGL((attn_hd256_kernel<WAVES>), dim3(a.qblocks * heads * nseg), dim3(64 * WAVES), 2 * G256_STAGE, st, a);
  return hwocr_launch_status();
}

template <int HD, bool CAUSAL>
int launch_prefill(const PrefillArgs& a, int nseg, int heads, int max_len, hipStream_t st) {
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int LDS = 2 * (64 * (HD * 2 + 16) + HDP * 144);
  HWOCR_PLAN("attn_prefill_kernel<%d,%s> nseg=%d heads=%d group=%d max_len=%d tiled=%d varlen=%d", HD, CAUSAL ? "causal" : "full", nseg,
             heads, a.group, max_len, a.kv_tiled, a.seg_off != nullptr);
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    hipFuncSetAttribute((const void*)attn_prefill_kernel<HD, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    return true;
  }();
  (void)done;
  dim3 grid((max_len + 127) / 128, heads, nseg), block(256);
  hipLaunchKernelGGL((attn_prefill_kernel<HD, CAUSAL>), grid, block, LDS, st, a);
  return hwocr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Vision-tower specialisation: head_dim 80, non-causal, row-major K / V^T (the hot attention of a page read:
// 16 heads x 5184 tokens x 32 layers per read).  Same mathematics as attn_prefill_kernel<80,false>, re-tiled for
// occupancy and VALU load:
//   * K / V^T tiles are staged by LDS-DMA (no staging registers, no ds_write): 3 workgroups per CU instead of 2.
//     LDS images are lane-linear per DMA instruction, so the bank swizzle is applied to the SOURCE address:
//       K  d 0..63  : [64 keys][128 B], chunk p of row r holds chunk p ^ ((r>>1)&7)
//       K  d 64..79 : [64 keys][ 32 B], chunk p of row r holds chunk p ^ ((r>>3)&1)
//       V^T         : [88 d   ][128 B], chunk p of row d holds chunk p ^ ((d>>1)&7); row 80 is all ones, so the
//                     padded third d-tile of the PV product accumulates the softmax denominator on the matrix pipe
//                     (the sum of the bf16-rounded weights, i.e. exactly the weights that multiply V) — 32 VALU adds
//                     per tile less, and the running-max rescale covers it for free.
//   * row max through v_max3 (fmaxf nests), exp2 of fma(score, scale, -max).
// ------------------------------------------------------------------------------------------------
// WAVES x 32 queries per workgroup.  Every workgroup streams the whole K / V^T of its (head, page) through LDS, so the
// bytes entering the CUs per query fall with the queries per workgroup: at 4 waves (128 queries, 3 workgroups per CU) a
// 12-page launch staged 13.5 GB in 1.8 ms = 7.5 TB/s, much of it from beyond L2 - the staging, not the matrix or vector pipes, was
// the bound.  12 waves (384 queries, one workgroup per CU, the same 3 waves per SIMD) stage a third of that.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 1) void attn_vit80_kernel(PrefillArgs a) {
  constexpr int HD = 80, NT = 64 * WAVES;
  constexpr int NSTG = 2;                   // LDS stages (a third one for the lone 12-wave workgroup measured slower: 3186 vs 3022 ms of vision per step)
  constexpr int DIST = NSTG - 1;            // K/V tiles in flight ahead of the one being multiplied
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages
  // 1-D grid of qblocks x (head, page) pairs.  Consecutive block ids go to different XCDs, so give every XCD whole
  // (head, page) pairs: all query blocks that re-read one K / V^T then share one L2 (measured before: 5x the unique
  // K/V bytes crossed the fabric).  Speed only.
  int seg, h, q0;
  {
    const int npairs = a.heads * a.nseg, L = blockIdx.x;
    int pair, qb;
    if ((npairs & 7) == 0) {
      const int slot = L >> 3;
      pair = (slot / a.qblocks) * 8 + (L & 7);
      qb = slot % a.qblocks;
    } else {
      pair = L / a.qblocks;
      qb = L % a.qblocks;
    }
    seg = pair / a.heads;
    h = pair % a.heads;
    q0 = qb * (32 * WAVES);
  }
  const int len = a.lens[seg];
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const bf16* Qp = a.Q + seg * a.q_seg + h * a.q_head;
  const bf16* Kp = a.K + seg * a.k_seg + h * a.k_head;
  const bf16* Vp = a.VT + seg * a.v_seg + h * a.v_head;
  const int qi = q0 + 32 * w + r;
  const int rk = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // tile row -> key permutation (bits 2,3 swapped)

  bf16x8 qf[5];
  {
    const bf16* qrow = Qp + (long)min(qi, len - 1) * a.q_row + 8 * hh;
#pragma unroll
    for (int s = 0; s < 5; ++s) qf[s] = *(const bf16x8*)(qrow + 16 * s);
  }
  // V^T rows 80..87 of both stages: row 80 = 1.0, the rest 0 (never touched by the DMA)
  for (int i = tid; i < NSTG * 8 * 64; i += NT) {
    const int stg = i >> 9, rr = (i >> 6) & 7, col = i & 63;
    ((bf16*)(smem + stg * V80_STAGE + V80_K0 + V80_K1 + (80 + rr) * 128))[col] = (bf16)(rr == 0 ? 1.0f : 0.0f);
  }

  // DMA plan per tile: 20 instructions of 1 KiB dealt round-robin to the waves: K d 0..63 (8: 8 key rows each),
  // K d 64..79 (2: 32 key rows each), V^T (10: 8 d rows each)
  const int nt = (len + 63) >> 6;
  auto stage_tile = [&](int t) {
    const int j0 = t * 64;
    char* st = smem + (t % NSTG) * V80_STAGE;
#pragma unroll
    for (int e = 0; e < (20 + WAVES - 1) / WAVES; ++e) {
      const int k = w + WAVES * e;  // wave-uniform
      if (k < 8) {
        const int row = 8 * k + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((const void*)(Kp + (long)min(j0 + row, len - 1) * a.k_row + lc * 8),
                                         LDS_PTR(st + k * 1024), 16, 0, 0);
      } else if (k < 10) {
        const int row = 32 * (k - 8) + (lane >> 1);
        const int lc = (lane & 1) ^ ((row >> 3) & 1);
        __builtin_amdgcn_global_load_lds((const void*)(Kp + (long)min(j0 + row, len - 1) * a.k_row + 64 + lc * 8),
                                         LDS_PTR(st + V80_K0 + (k - 8) * 1024), 16, 0, 0);
      } else if (k < 20) {
        const int d = 8 * (k - 10) + (lane >> 3);
        const int lc = (lane & 7) ^ ((d >> 1) & 7);
        __builtin_amdgcn_global_load_lds((const void*)(Vp + (long)d * a.v_row + j0 + lc * 8),
                                         LDS_PTR(st + V80_K0 + V80_K1 + (k - 10) * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 o[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = NEG_BIG;

  // this wave's DMA instructions per tile: tasks w, w + WAVES, .. < 20
  const int n_dma = (20 - w + WAVES - 1) / WAVES;
  auto wait_tiles_in_flight = [&](int tiles) {  // counted: leave `tiles` later tiles of this wave in flight
    const int pend = tiles * n_dma;
    if (pend <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (pend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // WAVES == 4: 5 per tile
  };
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nt) stage_tile(d);
  wait_tiles_in_flight(min(DIST - 1, nt - 1));
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * 64;
    char* st = smem + (t % NSTG) * V80_STAGE;
    if (t + DIST < nt) stage_tile(t + DIST);  // into the slot tile t-1 occupied (every wave passed the barrier after it)
    if (j0 + 64 > len) {
      // tail tile: keys past the segment must contribute 0 * finite.  Their scores are masked below; their V^T columns
      // are whatever the buffer holds, so clear them in LDS (one extra barrier, last tile only).
      for (int i = tid; i < 80 * 64; i += NT) {
        const int d = i >> 6, col = i & 63;
        if (j0 + col >= len) {
          const int ch = (col >> 3) ^ ((d >> 1) & 7);
          ((bf16*)(st + V80_K0 + V80_K1 + d * 128 + ch * 16))[col & 7] = (bf16)0.0f;
        }
      }
      __syncthreads();
    }
    const char* k0b = st;
    const char* k1b = st + V80_K0;
    const char* vb = st + V80_K0 + V80_K1;

    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ra = rk, rb = 32 + rk;
      const bf16x8 ka = *(const bf16x8*)(k0b + ra * 128 + (((2 * s + hh) ^ ((ra >> 1) & 7)) << 4));
      const bf16x8 kb2 = *(const bf16x8*)(k0b + rb * 128 + (((2 * s + hh) ^ ((rb >> 1) & 7)) << 4));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2, qf[s], s1, 0, 0, 0);
    }
    {
      const int ra = rk, rb = 32 + rk;
      const bf16x8 ka = *(const bf16x8*)(k1b + ra * 32 + ((hh ^ ((ra >> 3) & 1)) << 4));
      const bf16x8 kb2 = *(const bf16x8*)(k1b + rb * 32 + ((hh ^ ((rb >> 3) & 1)) << 4));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[4], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2, qf[4], s1, 0, 0, 0);
    }
    if (j0 + 64 > len) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = j0 + 16 * (i >> 3) + 8 * hh + (i & 7);
        if (key >= len) s0[i] = -INFINITY;
        if (key + 32 >= len) s1[i] = -INFINITY;
      }
    }
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s0[i]), s1[i]);  // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;
    if (__any(mx > m + a.slack)) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    bf16x8 pb[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[0][s2][j] = f2bf(__builtin_amdgcn_exp2f(__builtin_fmaf(s0[8 * s2 + j], a.scale_log2, -m)));
        pb[1][s2][j] = f2bf(__builtin_amdgcn_exp2f(__builtin_fmaf(s1[8 * s2 + j], a.scale_log2, -m)));
      }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int row = d * 32 + r;  // rows >= 88 of the last d-tile read past the V^T image: unused accumulator rows
          const bf16x8 vf = *(const bf16x8*)(vb + row * 128 + (((kt * 4 + s2 * 2 + hh) ^ ((row >> 1) & 7)) << 4));
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kt][s2], o[d], 0, 0, 0);
        }
    wait_tiles_in_flight(min(DIST - 1, nt - 2 - t));  // tile t+1 has landed
    __syncthreads();
  }

  // softmax denominator = accumulator row d = 80 (d-tile 2, local row 16 -> register 8 of the hh = 0 half)
  const float l = __shfl(o[2][8], r);
  const float inv = 1.0f / l;
  if (qi < len) {
    bf16* orow = a.O + seg * a.o_seg + (long)qi * a.o_row + h * HD;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * hh;
        if (dd < HD) {
          bf16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][4 * g + e] * inv);
          *(bf16x4*)(orow + dd) = ov;
        }
      }
  }
}

int launch_vit80(const PrefillArgs& a, int nseg, int heads, int max_len, hipStream_t st) {
  // long segments (pages: 5184 tokens): the one-wave-per-SIMD kernel of attention_vit80x.hip (256 queries per workgroup); short ones
  // keep 128-query workgroups so the grid still fills the chip.  HWOCR_VIT80_KERNEL = x | 12 | 4 forces a form (read per call, so a
  // test can walk all three in one process): 12 = the 384-query / 12-wave form that x replaced (2.13-2.21 ms against 2.07-2.12 per
  // 12-page launch; vision 2876-2940 -> 2837 ms per 84 pages in the bench).
  const char* force = getenv("HWOCR_VIT80_KERNEL");
  const bool wide = force ? force[0] == 'x' : max_len >= 1536;
  const int waves = force && force[0] != 'x' ? (atoi(force) == 12 ? 12 : 4) : (max_len >= 1536 ? 12 : 4);
  PrefillArgs b = a;
  b.heads = heads;
  b.nseg = nseg;
  HWOCR_PLAN("%s nseg=%d heads=%d max_len=%d", wide ? "attn_vit80x_kernel" : waves == 12 ? "attn_vit80_kernel<12>" : "attn_vit80_kernel<4>",
             nseg, heads, max_len);
  if (wide) {
    b.qblocks = (max_len + 255) / 256;
    b.slack = attn_slack();
    launch_vit80x(b, b.qblocks * heads * nseg, st);
    return hwocr_launch_status();
  }
  b.qblocks = (max_len + 32 * waves - 1) / (32 * waves);
  b.slack = attn_slack();
  if (waves == 12) {
    static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
      (void)hipFuncSetAttribute((const void*)attn_vit80_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * V80_STAGE);
      return true;
    }();
    (void)done;
    hipLaunchKernelGGL(attn_vit80_kernel<12>, dim3(b.qblocks * heads * nseg), dim3(768), 2 * V80_STAGE, st, b);
  } else {
    hipLaunchKernelGGL(attn_vit80_kernel<4>, dim3(b.qblocks * heads * nseg), dim3(256), 2 * V80_STAGE, st, b);
  }
  return hwocr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// decode attention
// ------------------------------------------------------------------------------------------------
struct DecodeArgs {
  const bf16* Q; const bf16* K; const bf16* VT; const int* lens;
  float* part_o; float* part_ml; bf16* out;
  long k_seq, k_head, v_seq, v_head, v_row;
  int Hq, Hkv, G, nsplit;
  float scale_log2;
  int kv_tiled;
  // fused QKV finish (hwocr_attn_decode_qkv; slabs == nullptr: Q, K, V^T were prepared by decode_qkv_finish_kernel): this
  // step's q / k / v arrive as the split-K slabs of the QKV projection
  const float* slabs = nullptr; int nslab = 0; long slab_stride = 0;
  const bf16* bias = nullptr; const int* rope_delta = nullptr; const bf16* cos_tab = nullptr; const bf16* sin_tab = nullptr;
  int ctx = 0, max_pos = 0; int* status = nullptr;
  bf16* Kw = nullptr; bf16* VTw = nullptr;  // the caches again, writable
  // nsplit > 1: one arrival counter per (read, kv head), zero between launches.  The workgroup that arrives last merges the
  // partials itself (nullptr: attn_decode_merge_kernel does, in a launch of its own)
  int* arrive = nullptr;
  int ahead = 1;  // fused QKV finish: first key block requested before the finish (HWOCR_ATTN_DECODE_AHEAD=0: after, for A/B runs)
  // E4M3 cache (hwocr_kv.fp8; the KV8 kernel instance): K / VT point at the CODES ([seq][Hkv][ctx * 256 bytes], kv8_k / kv8_v order),
  // one scale per token and kv head [seq][Hkv][ctx]
  float* k_scale = nullptr; float* v_scale = nullptr;
};

// partials of query head h of read b, output feature d: all <= 16 (m, l) pairs and output values are loaded up front (indices
// clamped, never a branch around a load: the loads of a head's 16 splits are one round trip to L2, not sixteen) and then combined
// in ascending split order - the same arithmetic whoever runs it (attn_decode_merge_kernel or the last workgroup of attn_decode_kernel)
// DEVICE_SCOPE: the partials were written by workgroups of the SAME launch on other XCDs - read them with device-scope loads (sc1:
// past this XCD's L2, which is not coherent with the others inside a launch)
template <int DEC_HD, bool DEVICE_SCOPE>
__device__ __forceinline__ void merge_splits(const DecodeArgs& a, int b, int h, int d) {
  auto ld = [](const float* p) {
    if constexpr (DEVICE_SCOPE) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
  };
  const int hk = h / a.G, qq = h - hk * a.G;
  const long base = ((long)b * a.Hkv + hk) * a.nsplit * a.G + qq;
  float m[16], l[16], o[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const long e = base + (long)(s < a.nsplit ? s : 0) * a.G;
    m[s] = ld(a.part_ml + e * 2);
    l[s] = ld(a.part_ml + e * 2 + 1);
    o[s] = ld(a.part_o + e * DEC_HD + d);
  }
  float M = NEG_BIG;
#pragma unroll
  for (int s = 0; s < 16; ++s) M = s < a.nsplit ? fmaxf(M, m[s]) : M;
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s < a.nsplit) {
      const float f = exp2f(m[s] - M);
      L += l[s] * f;
      O += o[s] * f;
    }
  }
  a.out[((long)b * a.Hq + h) * DEC_HD + d] = f2bf(O / L);
}


// Keys are walked in blocks of 32.  MFMA tile rows are assigned to keys so that the score registers a lane ends up with
// are 8 CONSECUTIVE keys (tile t, row 4a+r <-> key 8a + 4t + r): packed to bf16 they are the B operand of the PV product
// in natural k order, and the matching A operand is one 16-byte load of a V^T row.
// WAVES = 4: keys also split over `nsplit` workgroups (few reads in flight), partials merged by attn_decode_merge_kernel;
// WAVES = 8: one workgroup per (read, kv head) walks the whole cache and writes the final output - no merge launch.
// DEC_HD: 128, or 256 (Gemma; row layout only, 4 waves: the merge buffer is WAVES x DEC_HD x 16 floats).
// WPE: minimum waves per SIMD the register budget must allow.  The 8-wave form runs at its natural 145-159 VGPRs = one workgroup per
// CU; capped to 128 (WPE = 4: two workgroups per CU) hipcc spills 27 dwords and the decode step went from 4.21 to 4.94 ms per token.
// Keeping the NEXT block's 16 fragment loads in flight while the current block multiplies (double-buffered fragments: 256 VGPRs,
// 9 dwords of scratch) measured 4.32 against 4.22 ms per token: the loop is not waiting on its own loads.
// KV8 (round 4; head_dim 256, the fp8 configuration): the cache holds E4M3 codes in operand order (common.h kv8_k / kv8_v) + one
// scale per token: a block is 16 loads of 1 KiB per wave instead of 32, converted to bf16 fragments in registers (exact), the key
// scale multiplies the score, the value scale the softmax weight before it is packed; the appended token is quantised by the
// workgroup that owns its block.  Gemma's decode attention is 26 % of config 4's kernel time and streams KV at the HBM rate already
// (1.13 GB per layer at 5.4 TB/s): halving the bytes is what is left.
template <bool TILED, int WAVES, int DEC_HD, int WPE = (DEC_HD == 256 ? 1 : (WAVES == 8 ? 2 : 3)), bool KV8 = false>
__global__ __launch_bounds__(64 * WAVES, WPE) void attn_decode_kernel(DecodeArgs a) {
  constexpr int KS = DEC_HD / 32, VD = DEC_HD / 16;  // k-steps of the score product, d-tiles of the PV product
  static_assert(!TILED || DEC_HD == 128, "the fragment-tiled cache layout is defined for head_dim 128");
  static_assert(!KV8 || (DEC_HD == 256 && WAVES == 4 && !TILED), "the E4M3 cache is defined for 256-wide heads (4 waves = one thread per feature)");
  constexpr int QC = DEC_HD == 256 ? 8 : 16;  // query columns kept for the merge (64 KB static LDS limit): G <= QC
  __shared__ float s_o[WAVES][DEC_HD][QC];
  __shared__ float s_m[WAVES][16];
  __shared__ float s_l[WAVES][16];
  const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, qd = lane >> 4;
  const int len = a.lens[b];
  const bf16* Kp = a.K + b * a.k_seq + hk * a.k_head;
  const bf16* Vp = a.VT + b * a.v_seq + hk * a.v_head;
  // KV8: byte regions of ctx * DEC_HD codes, scales of ctx floats per (read, kv head)
  const long reg8 = ((long)b * a.Hkv + hk) * (long)a.ctx;
  const unsigned char* K8 = (const unsigned char*)a.K + reg8 * DEC_HD;
  const unsigned char* V8 = (const unsigned char*)a.VT + reg8 * DEC_HD;

  const int krow = 8 * (c >> 2) + (c & 3);  // key (within the block) of tile-0 row c; tile 1: +4
  const int nblk = (len + 31) >> 5;
  bf16x8 kf[KV8 ? 1 : 2][KV8 ? 1 : KS], vt[KV8 ? 1 : VD];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 kraw[KV8 ? 2 : 1][KV8 ? KS / 2 : 1], vraw[KV8 ? VD / 2 : 1];  // KV8: a lane's 16 codes of two k-steps / d-tiles per load
  f32x4 ksc[2], vsc[2];                                                // KV8: scales of this lane's 8 keys (tile t: keys 8 qd + 4 t + e)
  auto load_block = [&](int kb) {
    const int k0 = kb * 32;
    if constexpr (KV8) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s2 = 0; s2 < KS / 2; ++s2)
          kraw[t][s2] = __builtin_nontemporal_load((const i32x4*)(K8 + ((((long)kb * 2 + t) * (KS / 2) + s2) * 64 + lane) * 16));
#pragma unroll
      for (int d2 = 0; d2 < VD / 2; ++d2)
        vraw[d2] = __builtin_nontemporal_load((const i32x4*)(V8 + (((long)kb * (VD / 2) + d2) * 64 + lane) * 16));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ksc[t] = *(const f32x4*)(a.k_scale + reg8 + k0 + 8 * qd + 4 * t);
        vsc[t] = *(const f32x4*)(a.v_scale + reg8 + k0 + 8 * qd + 4 * t);
      }
    } else if constexpr (TILED) {  // one contiguous KiB per fragment, the cache is stored in operand order
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          kf[t][s] = __builtin_nontemporal_load((const bf16x8*)(Kp + ((((long)kb * 2 + t) * 4 + s) * 64 + lane) * 8));
#pragma unroll
      for (int d = 0; d < VD; ++d)
        vt[d] = __builtin_nontemporal_load((const bf16x8*)(Vp + (((long)kb * 8 + d) * 64 + lane) * 8));
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16* kr = Kp + (long)min(k0 + krow + 4 * t, len - 1) * DEC_HD + 8 * qd;
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[t][s] = *(const bf16x8*)(kr + 32 * s);
      }
#pragma unroll
      for (int d = 0; d < VD; ++d) vt[d] = *(const bf16x8*)(Vp + (long)(16 * d + c) * a.v_row + k0 + 8 * qd);
    }
  };
  // This wave's first key block is requested BEFORE the q / k / v finish below (3-4 us of slab sums, rotary and a cache append during
  // which nothing streamed) - unless it is the block that holds the slot being appended (the last one), which has to be read back after.
  int kb = split * WAVES + w;
  bool ahead = false;  // wave-uniform
  if (a.slabs && a.ahead && kb < nblk - 1 && len <= a.ctx) {  // (len > ctx: a broken host invariant - flagged below, nothing is read)
    load_block(kb);
    ahead = true;
  }

  bf16x8 qf[KS];
  if (a.slabs) {
    // ---- what decode_qkv_finish_kernel did in a launch of its own, for this (read, kv head): sum the split-K slabs of the
    // QKV projection (+ bias) of the G query heads, the key and the value head, round to bf16, rotate q and k at position
    // lens - 1 + rope_delta (the reference's bf16 rounding chain), append k / v to the cache at slot lens - 1 — then the
    // attention below reads the slot back like any other.  Scratch lives in s_o, which the loop does not touch.
    constexpr int HALF = DEC_HD / 2;
    bf16* s_row = (bf16*)&s_o[0][0][0];         // [G + 2][DEC_HD]: q heads, k, v as bf16(sum + bias)
    bf16* s_rot = s_row + 18 * DEC_HD;          // [G][DEC_HD]: rotated q heads
    const int slot = len - 1, pos = slot + a.rope_delta[b];
    if (slot < 0 || slot >= a.ctx || pos < 0 || pos >= a.max_pos) {  // uniform over every workgroup of this read
      // the host's invariants are broken (hwocr.h, HWOCR_STATUS_BAD_POSITION): nothing of this read is written
      if (tid == 0 && a.status) atomicOr(a.status, HWOCR_STATUS_BAD_POSITION);
      return;
    }
    // the workgroup whose key blocks include the new slot appends it (the only one that reads it back)
    const bool owner = split == ((slot >> 5) % (a.nsplit * WAVES)) / WAVES;
    const int W = (a.Hq + 2 * a.Hkv) * DEC_HD;
    for (int i4 = tid; i4 < (a.G + 2) * (DEC_HD / 4); i4 += 64 * WAVES) {
      const int which = i4 / (DEC_HD / 4), d = 4 * (i4 % (DEC_HD / 4));
      const int col = (which < a.G ? hk * a.G + which : which == a.G ? a.Hq + hk : a.Hq + a.Hkv + hk) * DEC_HD + d;
      f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int sl = 0; sl < a.nslab; ++sl) y += *(const f32x4*)(a.slabs + sl * a.slab_stride + (long)b * W + col);
      if (a.bias) {
        const bf16x4 bb = *(const bf16x4*)(a.bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] += bf2f(bb[e]);
      }
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = f2bf(y[e]);
      *(bf16x4*)(s_row + which * DEC_HD + d) = o4;
    }
    __syncthreads();
    for (int idx = tid; idx < (a.G + 1) * HALF; idx += 64 * WAVES) {
      const int which = idx / HALF, i = idx % HALF;
      if (which == a.G && !owner) continue;
      const float cs = bf2f(a.cos_tab[(long)pos * HALF + i]), sn = bf2f(a.sin_tab[(long)pos * HALF + i]);
      const float x1 = bf2f(s_row[which * DEC_HD + i]), x2 = bf2f(s_row[which * DEC_HD + HALF + i]);
      const bf16 oa = f2bf(rbf(x1 * cs) + rbf(-x2 * sn));
      const bf16 ob = f2bf(rbf(x2 * cs) + rbf(x1 * sn));
      if (which < a.G || KV8) {   // (KV8: the rotated key is staged behind the query heads and quantised below)
        s_rot[which * DEC_HD + i] = oa;
        s_rot[which * DEC_HD + HALF + i] = ob;
      } else {
        bf16* kb = a.Kw + b * a.k_seq + hk * a.k_head;
        kb[TILED ? kv_tiled_k(slot, i) : (long)slot * DEC_HD + i] = oa;
        kb[TILED ? kv_tiled_k(slot, HALF + i) : (long)slot * DEC_HD + HALF + i] = ob;
      }
    }
    if constexpr (KV8) {
      // the owner quantises the new token: thread d holds feature d of k and of v (256 threads, 256 features); scale = max|x| / 448
      // over the token's features (1 for an all-zero row), code = e4m3(x * 448 / max|x|), as hwocr_quant_rows_fp8
      __syncthreads();
      if (owner) {
        const float kx = bf2f(s_rot[a.G * DEC_HD + tid]), vx = bf2f(s_row[(a.G + 1) * DEC_HD + tid]);
        const float km = wave_max(fabsf(kx)), vm = wave_max(fabsf(vx));
        if (lane == 0) { s_m[w][0] = km; s_m[w][1] = vm; }
        __syncthreads();
        const float kmax = fmaxf(fmaxf(s_m[0][0], s_m[1][0]), fmaxf(s_m[2][0], s_m[3][0]));
        const float vmax = fmaxf(fmaxf(s_m[0][1], s_m[1][1]), fmaxf(s_m[2][1], s_m[3][1]));
        const float kinv = kmax > 0.f ? 448.0f / kmax : 0.f, vinv = vmax > 0.f ? 448.0f / vmax : 0.f;
        const int kc = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(kx * kinv, -448.0f), 448.0f), 0.f, 0, false);
        const int vc = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(vx * vinv, -448.0f), 448.0f), 0.f, 0, false);
        unsigned char* K8w = (unsigned char*)a.Kw + reg8 * DEC_HD;
        unsigned char* V8w = (unsigned char*)a.VTw + reg8 * DEC_HD;
        K8w[kv8_k(slot, tid)] = (unsigned char)(kc & 0xff);
        V8w[kv8_v(tid, slot)] = (unsigned char)(vc & 0xff);
        if (tid == 0) {
          a.k_scale[reg8 + slot] = kmax > 0.f ? kmax / 448.0f : 1.0f;
          a.v_scale[reg8 + slot] = vmax > 0.f ? vmax / 448.0f : 1.0f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the appended slot is in L2 before any wave reads it back
      }
    } else if (owner) {
      for (int d = tid; d < DEC_HD; d += 64 * WAVES)
        a.VTw[b * a.v_seq + hk * a.v_head + (TILED ? kv_tiled_v(d, slot) : (long)d * a.v_row + slot)] = s_row[(a.G + 1) * DEC_HD + d];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the appended slot is in L2 before any wave reads it back
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (c < a.G)
        qf[s] = *(const bf16x8*)(s_rot + c * DEC_HD + 32 * s + 8 * qd);
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (bf16)0.0f;
    }
    __syncthreads();  // s_o is the merge buffer again
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (c < a.G)
        qf[s] = *(const bf16x8*)(a.Q + ((long)b * a.Hq + hk * a.G + c) * DEC_HD + 32 * s + 8 * qd);
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (bf16)0.0f;
    }
  }

  f32x4 o[VD];
#pragma unroll
  for (int d = 0; d < VD; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_BIG, l = 0.f;

  for (; kb < nblk; kb += a.nsplit * WAVES) {
    const int k0 = kb * 32;
    if (!ahead) load_block(kb);
    ahead = false;
    if (k0 + 32 > len) {  // tail block: keys past the end carry p = 0, keep 0 * x finite
      if constexpr (KV8) {  // zero the codes (and the value scales) of this lane's keys past the end: bytes nv.. of both 8-byte halves
        const int nv = max(0, min(8, len - (k0 + 8 * qd)));
        const unsigned lo = nv >= 4 ? 0xffffffffu : (nv > 0 ? (1u << (8 * nv)) - 1u : 0u);
        const unsigned hi = nv >= 8 ? 0xffffffffu : (nv > 4 ? (1u << (8 * (nv - 4))) - 1u : 0u);
#pragma unroll
        for (int d2 = 0; d2 < VD / 2; ++d2) {
          vraw[d2].x &= (int)lo; vraw[d2].y &= (int)hi; vraw[d2].z &= (int)lo; vraw[d2].w &= (int)hi;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + 8 * qd + 4 * t + e >= len) vsc[t][e] = 0.f;
      } else {
#pragma unroll
        for (int d = 0; d < VD; ++d)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + 8 * qd + e >= len) vt[d][e] = (bf16)0.0f;
      }
    }
    // S^T tiles: lane (c, qd) register r of tile t <-> key k0 + 8qd + 4t + r, column = query c
    f32x4 sc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (KV8) {
#pragma unroll
        for (int s2 = 0; s2 < KS / 2; ++s2) {
          sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e4m3x8_bf16(kraw[t][s2].x, kraw[t][s2].y), qf[2 * s2], sc[t], 0, 0, 0);
          sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e4m3x8_bf16(kraw[t][s2].z, kraw[t][s2].w), qf[2 * s2 + 1], sc[t], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][s], qf[s], sc[t], 0, 0, 0);
      }
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = sc[t][e] * a.scale_log2;
        if constexpr (KV8) v *= ksc[t][e];   // the key's scale: q . (code * scale) = (q . code) * scale
        if (k0 + 8 * qd + 4 * t + e >= len) v = -INFINITY;
        sc[t][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float rs = 0.f;
    bf16x8 pb;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __builtin_amdgcn_exp2f(sc[t][e] - m_new);
        rs += pv;
        if constexpr (KV8) pb[4 * t + e] = f2bf(pv * vsc[t][e]);   // the value's scale rides on its softmax weight
        else pb[4 * t + e] = f2bf(pv);
      }
    l = l * alpha + rs;
    m = m_new;
    if constexpr (KV8) {
#pragma unroll
      for (int d2 = 0; d2 < VD / 2; ++d2) {
        o[2 * d2] *= alpha;
        o[2 * d2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e4m3x8_bf16(vraw[d2].x, vraw[d2].y), pb, o[2 * d2], 0, 0, 0);
        o[2 * d2 + 1] *= alpha;
        o[2 * d2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e4m3x8_bf16(vraw[d2].z, vraw[d2].w), pb, o[2 * d2 + 1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int d = 0; d < VD; ++d) {
        o[d] *= alpha;
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[d], pb, o[d], 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);

  // ---- merge the 4 waves through LDS; o[d][e] = O^T[16d + 4qd + e][query c]
#pragma unroll
  for (int d = 0; d < VD; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c < QC) s_o[w][16 * d + 4 * qd + e][c] = o[d][e];
  if (qd == 0) { s_m[w][c] = m; s_l[w][c] = l; }
  __syncthreads();
  for (int idx = tid; idx < a.G * DEC_HD; idx += 64 * WAVES) {
    const int qq = idx / DEC_HD, d = idx % DEC_HD;
    float M = s_m[0][qq];
#pragma unroll
    for (int ww = 1; ww < WAVES; ++ww) M = fmaxf(M, s_m[ww][qq]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) {
      const float f = __builtin_amdgcn_exp2f(s_m[ww][qq] - M);
      L += s_l[ww][qq] * f;
      O += s_o[ww][d][qq] * f;
    }
    if (a.nsplit == 1) {
      a.out[((long)b * a.Hq + hk * a.G + qq) * DEC_HD + d] = f2bf(O / L);
    } else {
      const long base = (((long)b * a.Hkv + hk) * a.nsplit + split) * a.G;
      if (a.arrive) {  // device-scope stores (sc1): visible to the merging workgroup on another XCD without flushing this XCD's L2
        __hip_atomic_store(a.part_o + (base + qq) * DEC_HD + d, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == 0) {
          __hip_atomic_store(a.part_ml + (base + qq) * 2, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.part_ml + (base + qq) * 2 + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        a.part_o[(base + qq) * DEC_HD + d] = O;
        if (d == 0) { a.part_ml[(base + qq) * 2] = M; a.part_ml[(base + qq) * 2 + 1] = L; }
      }
    }
  }
  if constexpr (WAVES == 4) {
    if (a.nsplit > 1 && a.arrive) {
      // The workgroup of this (read, kv head) that arrives LAST merges the partials of its G query heads: no second launch (a merge
      // launch was 6.7 us of a 3-read decode layer's 48).  The other splits ran on other XCDs, whose L2s are not coherent with this
      // one inside a launch: partials are stored and loaded at device scope (sc1), every wave has waited for its stores before the
      // barrier, and the count itself is a device-scope atomic.  (A __threadfence() per thread on either side - L2 write-back and
      // invalidate - measured SLOWER than the merge launch: 1.63 against 1.48 ms per 3-read token.)  The counter goes back to zero
      // for the next launch.
      __shared__ int s_last;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) s_last = __hip_atomic_fetch_add(a.arrive + b * a.Hkv + hk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.nsplit - 1;
      __syncthreads();
      if (!s_last) return;
      for (int idx = tid; idx < a.G * DEC_HD; idx += 64 * WAVES)
        merge_splits<DEC_HD, true>(a, b, hk * a.G + idx / DEC_HD, idx % DEC_HD);
      if (tid == 0) __hip_atomic_store(a.arrive + b * a.Hkv + hk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One workgroup per (query head, read), one thread per output feature (merge_splits above; the first form of this kernel walked
// the splits in a loop of dependent loads from 6 workgroups and took 22 us of a 3-read decode layer's 70, r03c profile).  Launched
// only when the caller gives attn_decode_kernel no arrival counters.
template <int DEC_HD>
__global__ __launch_bounds__(DEC_HD) void attn_decode_merge_kernel(DecodeArgs a) {
  merge_splits<DEC_HD, false>(a, blockIdx.y, blockIdx.x, threadIdx.x);
}

}  // namespace

extern "C" int hwocr_attn_prefill(const void* Q, const void* K, const void* VT, void* O, const int* lens,
                                  int nseg, int heads, int group, int head_dim, int max_len, int causal,
                                  long q_seg, long q_head, long q_row, long k_seg, long k_head, long k_row,
                                  long v_seg, long v_head, long v_row, long o_seg, long o_row, float scale,
                                  int kv_tiled, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (nseg <= 0 || heads <= 0 || group <= 0 || max_len <= 0) return HWOCR_EINVAL;
  if ((q_row % 8) || (k_row % 8) || (v_row % 8) || (o_row % 4) || (q_head % 8) || (k_head % 8) || (v_head % 8) ||
      (q_seg % 8) || (k_seg % 8) || (v_seg % 8))
    return HWOCR_EINVAL;
  PrefillArgs a{(const bf16*)Q, (const bf16*)K, (const bf16*)VT, (bf16*)O, lens,
                q_seg, q_head, q_row, k_seg, k_head, k_row, v_seg, v_head, v_row, o_seg, o_row,
                group, scale * 1.4426950408889634f, kv_tiled, heads, nseg, (max_len + 127) / 128};
  if (kv_tiled && head_dim != 128) return HWOCR_EINVAL;
  if (head_dim == 80 && !causal && group == 1) {
    static const bool generic = HWOCR_DIAG_ENV_INT("HWOCR_ATTN_GENERIC", 0) != 0;
    if (!generic) return launch_vit80(a, nseg, heads, max_len, stream);
  }
  if (head_dim == 80) return causal ? launch_prefill<80, true>(a, nseg, heads, max_len, stream)
                                    : launch_prefill<80, false>(a, nseg, heads, max_len, stream);
  if (head_dim == 128) return causal ? launch_prefill<128, true>(a, nseg, heads, max_len, stream)
                                     : launch_prefill<128, false>(a, nseg, heads, max_len, stream);
  if (head_dim == 256 && !causal && !kv_tiled) {
    // HWOCR_HD256_WAVES=0: generic kernel.  (8 waves x 32 queries would halve the staged bytes per query but needs
    // 256 + VGPRs per wave: hipcc spills 74 of them and the kernel measured no faster than the generic one.  32-key tiles
    // in four LDS stages - three tiles in flight instead of one - measured 4.1-4.3 ms at 4 waves and 3.6-4.2 ms at 8 (21-35
    // spills) against 3.47 ms: the time is not DMA latency.)
    static const int waves = HWOCR_DIAG_ENV_INT("HWOCR_HD256_WAVES", 4);
    if (waves == 4) return launch_hd256<4>(a, nseg, heads, max_len, stream);
  }
  if (head_dim == 256) return causal ? launch_prefill<256, true>(a, nseg, heads, max_len, stream)
                                     : launch_prefill<256, false>(a, nseg, heads, max_len, stream);
  if (head_dim == 64) return causal ? launch_prefill<64, true>(a, nseg, heads, max_len, stream)
                                    : launch_prefill<64, false>(a, nseg, heads, max_len, stream);
  if (head_dim == 32) return causal ? launch_prefill<32, true>(a, nseg, heads, max_len, stream)
                                    : launch_prefill<32, false>(a, nseg, heads, max_len, stream);
  return HWOCR_EINVAL;
}

extern "C" int hwocr_attn_varlen(const void* Q, const void* K, const void* VT, void* O, const int* seg_off,
                                 const int* lens, int nseg, int heads, int head_dim, int max_len, long q_head,
                                 long q_row, long k_head, long k_row, long v_head, long v_row, long o_row, float scale,
                                 hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!seg_off || !lens || nseg <= 0 || heads <= 0 || max_len <= 0) return HWOCR_EINVAL;
  if ((q_row % 8) || (k_row % 8) || (v_row % 8) || (o_row % 4) || (q_head % 8) || (k_head % 8) || (v_head % 8))
    return HWOCR_EINVAL;
  PrefillArgs a{(const bf16*)Q, (const bf16*)K, (const bf16*)VT, (bf16*)O, lens,
                0, q_head, q_row, 0, k_head, k_row, 0, v_head, v_row, 0, o_row,
                1, scale * 1.4426950408889634f, 0, heads, nseg, (max_len + 127) / 128, seg_off};
  switch (head_dim) {
    case 32: return launch_prefill<32, false>(a, nseg, heads, max_len, stream);
    case 64: return launch_prefill<64, false>(a, nseg, heads, max_len, stream);
    case 80: return launch_prefill<80, false>(a, nseg, heads, max_len, stream);
    case 128: return launch_prefill<128, false>(a, nseg, heads, max_len, stream);
  }
  return HWOCR_EINVAL;
}

namespace {
int launch_attn_decode(const DecodeArgs& a, int nseq, int head_dim, hipStream_t stream) {
  const int Hkv = a.Hkv, nsplit = a.nsplit;
  HWOCR_PLAN("attn_decode_kernel<%s,%d,%d>%s fused_qkv=%d nseq=%d Hq=%d Hkv=%d nsplit=%d nslab=%d", a.k_scale ? "e4m3" : a.kv_tiled ? "tiled" : "rows",
             (head_dim == 256 || nsplit > 1) ? 4 : 8, head_dim, nsplit > 1 ? (a.arrive ? "+lastwg" : "+merge") : "", a.slabs != nullptr, nseq, a.Hq, Hkv, nsplit, a.nslab);
  if (head_dim == 256) {  // 4 waves; a single pass when the caller asks for no split
    if (a.k_scale) hipLaunchKernelGGL((attn_decode_kernel<false, 4, 256, 1, true>), dim3(nsplit, Hkv, nseq), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_decode_kernel<false, 4, 256>), dim3(nsplit, Hkv, nseq), dim3(256), 0, stream, a);
    if (nsplit > 1 && !a.arrive) hipLaunchKernelGGL(attn_decode_merge_kernel<256>, dim3(a.Hq, nseq), dim3(256), 0, stream, a);
    return hwocr_launch_status();
  }
  if (nsplit == 1) {
    if (a.kv_tiled) hipLaunchKernelGGL((attn_decode_kernel<true, 8, 128>), dim3(1, Hkv, nseq), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((attn_decode_kernel<false, 8, 128>), dim3(1, Hkv, nseq), dim3(512), 0, stream, a);
  } else if (a.kv_tiled) {
    hipLaunchKernelGGL((attn_decode_kernel<true, 4, 128>), dim3(nsplit, Hkv, nseq), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<false, 4, 128>), dim3(nsplit, Hkv, nseq), dim3(256), 0, stream, a);
  }
  if (nsplit > 1 && !a.arrive) hipLaunchKernelGGL(attn_decode_merge_kernel<128>, dim3(a.Hq, nseq), dim3(128), 0, stream, a);
  return hwocr_launch_status();
}
bool attn_decode_args_ok(int nseq, int Hq, int Hkv, int nsplit, float* part_o, float* part_ml, long k_seq, long k_head,
                         long v_seq, long v_head, long v_row, int head_dim, int kv_tiled) {
  // nsplit <= 16: the partial buffers of hwocr_dec_ws are sized for 16 splits (engine._dec_ws)
  if (nseq <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || Hq / Hkv > 16 || nsplit < 1 || nsplit > 16) return false;
  if (nsplit > 1 && (!part_o || !part_ml)) return false;
  if ((v_row % 64) || (k_seq % 8) || (k_head % 8) || (v_seq % 8) || (v_head % 8)) return false;
  if ((head_dim != 128 && head_dim != 256) || (kv_tiled && head_dim != 128)) return false;
  if (head_dim == 256 && Hq / Hkv > 8) return false;
  return true;
}
}  // namespace

extern "C" int hwocr_attn_decode(const void* Q, const void* K, const void* VT, const int* lens, void* out,
                                 float* part_o, float* part_ml, int* arrive, int nseq, int Hq, int Hkv, int nsplit,
                                 long k_seq, long k_head, long v_seq, long v_head, long v_row, float scale,
                                 int head_dim, int kv_tiled, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!attn_decode_args_ok(nseq, Hq, Hkv, nsplit, part_o, part_ml, k_seq, k_head, v_seq, v_head, v_row, head_dim, kv_tiled))
    return HWOCR_EINVAL;
  DecodeArgs a{(const bf16*)Q, (const bf16*)K, (const bf16*)VT, lens, part_o, part_ml, (bf16*)out,
               k_seq, k_head, v_seq, v_head, v_row, Hq, Hkv, Hq / Hkv, nsplit, scale * 1.4426950408889634f, kv_tiled};
  a.arrive = arrive;
  return launch_attn_decode(a, nseq, head_dim, stream);
}

// hwocr_decode_qkv_finish + hwocr_attn_decode in one launch (hwocr.h)
extern "C" int hwocr_attn_decode_qkv(const float* slabs, int nslab, long slab_stride, const void* bias, void* K, void* VT,
                                     const int* lens, const int* rope_delta, const void* cos_tab, const void* sin_tab, void* out,
                                     float* part_o, float* part_ml, int* arrive, int nseq, int Hq, int Hkv, int nsplit, long k_seq,
                                     long k_head, long v_seq, long v_head, long v_row, float scale, int head_dim, int kv_tiled,
                                     int ctx, int max_pos, int* status, hipStream_t stream) {
  (void)hipGetLastError();
  if (!attn_decode_args_ok(nseq, Hq, Hkv, nsplit, part_o, part_ml, k_seq, k_head, v_seq, v_head, v_row, head_dim, kv_tiled))
    return HWOCR_EINVAL;
  if (!slabs || nslab < 1 || !rope_delta || !cos_tab || !sin_tab || ctx < 1 || max_pos < 1 || !K || !VT) return HWOCR_EINVAL;
  DecodeArgs a{nullptr, (const bf16*)K, (const bf16*)VT, lens, part_o, part_ml, (bf16*)out,
               k_seq, k_head, v_seq, v_head, v_row, Hq, Hkv, Hq / Hkv, nsplit, scale * 1.4426950408889634f, kv_tiled};
  a.slabs = slabs; a.nslab = nslab; a.slab_stride = slab_stride; a.bias = (const bf16*)bias; a.rope_delta = rope_delta;
  a.cos_tab = (const bf16*)cos_tab; a.sin_tab = (const bf16*)sin_tab; a.ctx = ctx; a.max_pos = max_pos; a.status = status;
  a.Kw = (bf16*)K; a.VTw = (bf16*)VT; a.arrive = arrive;
  static const int ahead = HWOCR_DIAG_ENV_INT("HWOCR_ATTN_DECODE_AHEAD", 1);
  a.ahead = ahead;
  return launch_attn_decode(a, nseq, head_dim, stream);
}

// the fused step over an E4M3 cache (hwocr.h)
extern "C" int hwocr_attn_decode_qkv_fp8kv(const float* slabs, int nslab, long slab_stride, const void* bias, void* K8, void* VT8,
                                           float* k_scale, float* v_scale, const int* lens, const int* rope_delta, const void* cos_tab,
                                           const void* sin_tab, void* out, float* part_o, float* part_ml, int* arrive, int nseq, int Hq,
                                           int Hkv, int nsplit, float scale, int ctx, int max_pos, int* status, hipStream_t stream) {
  (void)hipGetLastError();
  if (!attn_decode_args_ok(nseq, Hq, Hkv, nsplit, part_o, part_ml, 0, 0, 0, 0, 0, 256, 0)) return HWOCR_EINVAL;
  if (!slabs || nslab < 1 || !rope_delta || !cos_tab || !sin_tab || ctx < 32 || (ctx % 32) || max_pos < 1 || !K8 || !VT8 || !k_scale || !v_scale)
    return HWOCR_EINVAL;
  DecodeArgs a{nullptr, (const bf16*)K8, (const bf16*)VT8, lens, part_o, part_ml, (bf16*)out, 0, 0, 0, 0, 0, Hq, Hkv, Hq / Hkv, nsplit,
               scale * 1.4426950408889634f, 0};
  a.slabs = slabs; a.nslab = nslab; a.slab_stride = slab_stride; a.bias = (const bf16*)bias; a.rope_delta = rope_delta;
  a.cos_tab = (const bf16*)cos_tab; a.sin_tab = (const bf16*)sin_tab; a.ctx = ctx; a.max_pos = max_pos; a.status = status;
  a.Kw = (bf16*)K8; a.VTw = (bf16*)VT8; a.arrive = arrive; a.k_scale = k_scale; a.v_scale = v_scale;
  a.ahead = 1;
  return launch_attn_decode(a, nseq, 256, stream);
}

// the kernel instance hwocr_attn_decode runs for these arguments (for the parity tests' coverage check)
extern "C" int hwocr_attn_decode_variant(int nsplit, int head_dim, int kv_tiled, char* name, int name_len) {
  if (!name || name_len < 8 || nsplit < 1 || nsplit > 16 || (head_dim != 128 && head_dim != 256) || (kv_tiled && head_dim != 128))
    return HWOCR_EINVAL;
  const int waves = (head_dim == 256 || nsplit > 1) ? 4 : 8;
  snprintf(name, name_len, "attn_decode_kernel<%s,%d,%d>%s", kv_tiled ? "tiled" : "rows", waves, head_dim, nsplit > 1 ? "+merge" : "");
  return HWOCR_OK;
}
