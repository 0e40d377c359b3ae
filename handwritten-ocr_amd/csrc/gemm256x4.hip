// 256x256 bf16 GEMM, second structure: FOUR waves of 128 x 128 per workgroup (one wave per SIMD, 512 registers each).
//
// Why: the ablation of the 8-wave kernel (gemm256.hip) shows that neither the matrix pipe (84 % of peak
// alone) nor the staging path (13-17 TB/s alone) holds it at 45-56 %, but the LDS fragment reads (-19 %: one of a SIMD's two
// waves fetches while the other multiplies for about one LDS latency) and the DMA fills of the same LDS (-17 %).  With the
// 512-register budget of a lone wave per SIMD:
//   * a wave owns 128 x 128 outputs = 8 x 8 MFMA 16x16x32 tiles (256 accumulator registers): (128 + 128) fragment rows per
//     16384 outputs instead of (128 + 64) per 8192 - a third fewer LDS bytes per FLOP;
//   * the fragments of K tile t+1 (16 x ds_read_b128, 64 VGPRs) are fetched at the START of the 64 MFMAs of K tile t into a
//     second register buffer: a whole multiply interval (1024 matrix-pipe cycles) covers the LDS latency;
//   * K is walked in 32-wide tiles: a stage = 256 activation rows + 256 weight rows of 64 B = 32 KiB, FOUR stages; the tile
//     whose fragments have just been read is refilled for K tile t+5, so three to four tiles are in flight (about 1.7 us
//     between the DMA issue and the first read of its bytes); one barrier per K tile.
// LDS image: rows of 64 B, 16-byte chunk p of row r holds chunk p ^ ((r >> 2) & 3) (16 consecutive rows x one chunk =
// 16 different 16-byte slots of a 256-byte bank row); filled by LDS-DMA with the permutation on the source address.
// The weight tile is the MFMA A operand (a lane owns 4 consecutive output features), as in the 8-wave kernel.
#include "gemm_common.cuh"
#include <cstdlib>
#include <type_traits>

using namespace gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int HALF = 256 * BK * 2;   // 16 KiB: one operand's rows of a K tile
constexpr int STAGE = 2 * HALF;      // activation rows, then weight rows
constexpr int NSTG = 4;
constexpr int LDS_BYTES = NSTG * STAGE;  // 128 KiB
constexpr int DMA_PER_TILE = 8;      // 1-KiB instructions per wave per K tile (32 in all)

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  switch (tiles) {
    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_wide256x4_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  int tm, tn;
  tile_of_block(a.tilesM, a.tilesN, 4, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = a.K / BK;

  // ---- staging plan: instruction j of a wave copies 16 tile rows (1 KiB, lane-linear in LDS): j < 4 activation rows
  // 64 w + 16 j .., j >= 4 weight rows 64 w + 16 (j - 4) ..; the lane's chunk carries the swizzle on the SOURCE side
  const int lrow = lane >> 2, pc = lane & 3;
  const char* src[DMA_PER_TILE];
#pragma unroll
  for (int j = 0; j < DMA_PER_TILE; ++j) {
    const int row = 64 * w + 16 * (j & 3) + lrow;
    const int lc = pc ^ ((row >> 2) & 3);
    src[j] = j < 4 ? (const char*)(a.X + (size_t)min(m0 + row, a.M - 1) * a.ldx) + lc * 16
                   : (const char*)(a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw) + lc * 16;
  }
  auto issue = [&](int t) {  // K tile t -> stage t % 4
    char* st = smem + (t & (NSTG - 1)) * STAGE + (64 * w) * 64;
#pragma unroll
    for (int j = 0; j < DMA_PER_TILE; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(src[j] + (size_t)t * (BK * 2)),
                                       LDS_PTR(st + (j < 4 ? 0 : HALF) + (j & 3) * 1024), 16, 0, 0);
  };

  // ---- fragments: lane (c, q) of tile i reads row base + 16 i + c, logical chunk q
  // (the swizzle of row 128 wr + 16 i + c is (c >> 2) & 3 for every i)
  const int chunk = (q ^ ((c >> 2) & 3)) << 4;
  const int xoff = (128 * wr + c) * 64 + chunk;          // + 16 i * 64
  const int woff = HALF + (128 * wc + c) * 64 + chunk;   // + 16 j * 64
  i32x4 xf[2][8], wf[2][8];
  // The reads are inline asm as well: hipcc does not track them, so it cannot put one of its conservative
  // "s_waitcnt lgkmcnt(0)" between a step's fresh reads and the MFMAs that do not depend on them (it did, in every second
  // step); the explicit wait in the middle of a step is the only one, and the destination registers are touched by nothing
  // but the next step's asm MFMAs.
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
  auto read_frags = [&xf, &wf, lds0, xoff, woff](int t, auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
    const unsigned xa = lds0 + (t & (NSTG - 1)) * STAGE + xoff, wa = lds0 + (t & (NSTG - 1)) * STAGE + woff;
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[buf][i]) : "v"(xa), "n"(i * 1024));
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[buf][j]) : "v"(wa), "n"(j * 1024));
  };

  f32x4 acc[8][8];  // [weight tile j][activation tile i]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mma_half = [&acc, &xf, &wf](auto buf_c, int jh) {  // weight tiles 4 jh .. 4 jh + 3 against all activation tiles: 32 MFMAs
    constexpr int buf = decltype(buf_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 4 * jh; j < 4 * jh + 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        // inline asm pins the register classes: the 256 accumulator registers ARE the AGPR file, everything else lives in
        // VGPRs (left to itself hipcc splits both sets across the two files and copies 236 registers around the loop)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[buf][j]), "v"(xf[buf][i]));
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: K tiles 0..3 on their way, tiles 0 and 1 landed, fragments of tile 0 in buffer 0, stage 0 refilled
  const int pre = min(nk, NSTG);
  for (int t = 0; t < pre; ++t) issue(t);
  int issued = pre;
  wait_tiles_in_flight(max(0, issued - 1));  // tile 0
  __builtin_amdgcn_s_barrier();
  read_frags(0, std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(xf[0][i]), "v"(wf[0][i]));  // (hipcc's wait belongs here, see step)
  wait_tiles_in_flight(max(0, issued - 2));  // tile 1
  __builtin_amdgcn_s_barrier();               // every wave has read stage 0, tile 1 is complete
  if (issued < nk) { issue(issued); ++issued; }  // K tile 4 -> stage 0

  // one K tile; CUR = register buffer holding its fragments (K/64 is whole, so tiles come in (even, odd) pairs: no
  // run-time buffer choice, no register copies)
  auto step = [&](int t, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    // fragments of K tile t+1 start their way into the other buffer; they land under the first 32 MFMAs
    // (unconditional, so no control-flow join - and no conservative wait - sits between these reads and the MFMAs: the
    // last step re-reads its own stage, which nobody refills any more)
    read_frags(min(t + 1, nk - 1), std::integral_constant<int, CUR ^ 1>{});
    __builtin_amdgcn_sched_barrier(0);
    mma_half(cur_c, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) {
      // own fragment reads of stage (t+1) % 4 are done.  The empty asm "uses" the new fragments here, so hipcc's own
      // lgkmcnt wait lands at this point and not in front of the next step's first MFMA, behind that step's fresh reads
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(xf[CUR ^ 1][i]), "v"(wf[CUR ^ 1][i]));
      // K tile t+2 must be complete before the next step reads it: leave only the tiles issued after it in flight
      wait_tiles_in_flight(max(0, min(issued - 1 - (t + 2), 3)));
      __builtin_amdgcn_s_barrier();  // (a) every wave is done with stage (t+1) % 4, (b) every wave's share of tile t+2 landed
      if (issued < nk) { issue(issued); ++issued; }  // K tile t+5 -> stage (t+1) % 4
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_half(cur_c, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, std::integral_constant<int, 0>{});
    step(t + 1, std::integral_constant<int, 1>{});
  }
  // the asm MFMAs are opaque to the hazard recogniser: let the last ones leave the matrix pipe before accumulators are read
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: lane (c,q) of tile (j,i) holds out[m0 + 128wr + 16i + c][n0 + 128wc + 16j + 4q .. +3]
  if constexpr (is_glu<EPI>) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + 128 * wr + 16 * i + c;
#pragma unroll
      for (int j = 0; j < 8; j += 2) store_glu<EPI>(a, acc[j][i], acc[j + 1][i], m, n0 + 128 * wc + 16 * j, q);
    }
  } else {
    // through LDS so that HBM sees whole 256-byte rows: a private 32 KiB per wave = one (idle) stage; the barrier makes
    // sure nobody still reads fragments of the last K tiles from it
    __builtin_amdgcn_s_barrier();
    char* ep = smem + w * STAGE;  // [128 rows][256 B], 16-byte chunk p of row r at p ^ (r & 15)
    bf16x4 bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + 128 * wc + 16 * j + 4 * q;
      bv[j] = (a.bias && n < a.N) ? *(const bf16x4*)(a.bias + n) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ml = 16 * i + c;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][i][r] + bf2f(bv[j][r]);
          if constexpr (EPI == EPI_QUICKGELU) v = act_quick_gelu(rbf(v));
          else if constexpr (EPI == EPI_GELU) v = act_gelu_erf(rbf(v));
          else if constexpr (EPI == EPI_GELU_TANH) v = act_gelu_tanh(rbf(v));
          o[r] = f2bf(v);
        }
        *(bf16x4*)(ep + ml * 256 + (((2 * j + (q >> 1)) ^ (ml & 15)) << 4) + (q & 1) * 8) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own wave's writes are in LDS before any lane reads them
    __builtin_amdgcn_wave_barrier();
    const int pch = lane & 15;
    const int n = n0 + 128 * wc + 8 * pch;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
      const int row = 4 * i + (lane >> 4);
      const int m = m0 + 128 * wr + row;
      bf16x8 v = *(const bf16x8*)(ep + row * 256 + ((pch ^ (row & 15)) << 4));
      if (m < a.M && n < a.N) {
        if constexpr (EPI == EPI_RESIDUAL) {
          const bf16x8 rs = *(const bf16x8*)(a.res + (size_t)m * a.ldres + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rs[e]));
        }
        *(bf16x8*)(a.out + (size_t)m * a.ldo + n) = v;
      }
    }
  }
}

template <int EPI>
void launch(const WideArgs& a, hipStream_t st) {
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void*)gemm_wide256x4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    done = true;
  }
  WideArgs b = a;
  b.tilesM = (a.M + BM - 1) / BM;
  b.tilesN = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_wide256x4_kernel<EPI>), dim3(b.tilesM * b.tilesN), dim3(256), LDS_BYTES, st, b);
}

}  // namespace

int hwocr_gemm_wide256x4(const WideArgs& a, int epi, hipStream_t stream) {
  if (a.K % (2 * BK) || a.K < 2 * BK) return HWOCR_EINVAL;
  switch (epi) {
    case EPI_LINEAR: launch<EPI_LINEAR>(a, stream); break;
    case EPI_RESIDUAL: launch<EPI_RESIDUAL>(a, stream); break;
    case EPI_QUICKGELU: launch<EPI_QUICKGELU>(a, stream); break;
    case EPI_GELU: launch<EPI_GELU>(a, stream); break;
    case EPI_SWIGLU: launch<EPI_SWIGLU>(a, stream); break;
    case EPI_GELU_TANH: launch<EPI_GELU_TANH>(a, stream); break;
    case EPI_GEGLU: launch<EPI_GEGLU>(a, stream); break;
    default: return HWOCR_EINVAL;
  }
  return hwocr_launch_status();
}
