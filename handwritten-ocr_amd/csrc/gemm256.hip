// 256x256x64 bf16 GEMM for the MFMA-bound part of a page read (vision tower, merger, decoder prefill).
//
// Why a second wide kernel: at 128x128 tiles every MFMA FLOP pulls twice the operand bytes L2 -> LDS; at full MFMA rate
// that is 64 B/clk/CU, the whole L2->CU path.  A 256x256 tile halves it and one 512-thread workgroup owns the CU.
//
// Structure (one workgroup = 8 waves = 2(M) x 4(N), each wave a 128 x 64 output block = 8 x 4 MFMA 16x16x32 tiles):
//   * LDS: 2 stages x (activation tile 256 x 64 + weight tile 256 x 64) bf16 = 128 KiB, rows of 128 B with the 16-byte
//     chunks XOR-swizzled (conflict-free ds_read_b128), filled by LDS-DMA with the permutation on the SOURCE address.
//   * A K tile is staged as 4 units of 16 KiB (2 DMA instructions per thread each), ordered as they are consumed:
//       U0 = activation rows of both wave-rows' first 64-row half, U1 = weight rows of every wave-column's first
//       32-row half, U2 = second weight halves, U3 = second activation halves.
//   * A K tile is multiplied in 4 phases of 16 MFMAs per wave (one quadrant of the wave's block each):
//       ph0 reads U0,U1 -> (m0,n0);  ph1 reads U2 -> (m0,n1);  ph2 reads U3 -> (m1,n1);  ph3 reads nothing -> (m1,n0).
//   * Every phase issues ONE unit, 7 units (1.75 K tiles) ahead of the consumer; the unit it overwrites was last read one
//     phase earlier.  Only ph3 waits, with a COUNTED vmcnt that retires the next K tile and leaves 3 units in flight.
// The weight tile is the MFMA A operand, so each lane owns 4 consecutive output features (8-byte stores).
//
// Persistent tile loop (round 2).  The grid is one workgroup per CU; workgroup b walks output tiles b, b + grid, ... .  With one
// tile per workgroup (round 1) a CU spent, per 45 us tile of the K = 1280 tower GEMMs, ~1.5 us waiting for the next workgroup
// to be dispatched and ~2 us for its first K tile to arrive, with the matrix pipe idle.  Now, as soon as the last K tile of a
// tile has been multiplied, the 7 prologue units of the NEXT tile are issued, and only then does the epilogue of the finished
// tile run (bias + activation in registers, staged through a 4-KiB-per-wave LDS region of its own — the operand stages belong to
// the prologue DMAs by then — and stored as whole 128-byte rows): the first K tiles land under the epilogue, and the epilogue's
// stores drain under the next tile's first K tiles (counted wait between two tiles, see the end of the loop).
//
// FP8 variant (BASELINE config 4: E4M3 weights + activations): the SAME byte-level pipeline over rows of 128 fp8 values
// (a K tile = 128 elements, so half as many K tiles), multiplied with v_mfma_f32_16x16x128_f8f6f4 (the 2x-rate form): a
// lane's 32-byte operand = the two 16-byte chunks the bf16 form feeds to its two k-steps.  The k order inside the
// instruction is therefore permuted, identically for both operands, which a dot product does not see.  Operands carry
// one fp32 scale per row (activation) / per output feature (weight); the epilogue multiplies them in before the bias.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

using namespace gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE = 256 * BK * 2;   // 32 KiB per operand tile
constexpr int STAGE = 2 * TILE;      // activation tile, then weight tile
constexpr int EPI_STAGE = 8 * 4096;  // epilogue staging: 4 KiB per wave, apart from the operand stages (see the tile loop)
constexpr int LDS_BYTES = 2 * STAGE + EPI_STAGE;  // 128 + 32 = 160 KiB: the whole LDS of a CU
constexpr int AHEAD = 7;             // units in flight ahead of the consuming phase

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <bool FP8, int MH, int NH, bool WIDE32 = false>
__device__ __forceinline__ void quadrant_mma(f32x4 (&acc)[4][8], const i32x4 (&wf)[2][2][2], const i32x4 (&xf)[4][2]) {
  __builtin_amdgcn_s_setprio(1);
  if constexpr (WIDE32) {
    // diagnostic (HWOCR_GEMM_ABLATE=11, WRONG results): the same fragments through HALF as many v_mfma_f32_32x32x16_bf16 - the same
    // matrix-pipe cycles and LDS traffic, but the pipe holds the SIMD's issue port 8 cycles in 32 instead of 8 in 16
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4(&r)[8] = acc[2 * NH + j];
        f32x16_t c = __builtin_shufflevector(__builtin_shufflevector(r[4 * MH], r[4 * MH + 1], 0, 1, 2, 3, 4, 5, 6, 7),
                                             __builtin_shufflevector(r[4 * MH + 2], r[4 * MH + 3], 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5,
                                             6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[NH][j][kk]), __builtin_bit_cast(bf16x8, xf[2 * i + j][kk]), c,
                                                      0, 0, 0);
        r[4 * MH] = __builtin_shufflevector(c, c, 0, 1, 2, 3);
        r[4 * MH + 1] = __builtin_shufflevector(c, c, 4, 5, 6, 7);
        r[4 * MH + 2] = __builtin_shufflevector(c, c, 8, 9, 10, 11);
        r[4 * MH + 3] = __builtin_shufflevector(c, c, 12, 13, 14, 15);
      }
  } else if constexpr (FP8) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x8 wa = __builtin_shufflevector(wf[NH][j][0], wf[NH][j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 xb = __builtin_shufflevector(xf[i][0], xf[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
        // formats 0 = E4M3 for both operands; scale operands 0 select the unscaled encoding (block scales of 1)
        acc[2 * NH + j][4 * MH + i] =
            __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[2 * NH + j][4 * MH + i], 0, 0, 0, 0, 0, 0);
      }
  } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[2 * NH + j][4 * MH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, wf[NH][j][kk]), __builtin_bit_cast(bf16x8, xf[i][kk]),
              acc[2 * NH + j][4 * MH + i], 0, 0, 0);
  }
  __builtin_amdgcn_s_setprio(0);
}

// ablation (tools/bench_gemm_ablate.py): fragments stay live without the matrix pipe
__device__ __forceinline__ void keep_alive(f32x4 (&acc)[4][8], const i32x4 (&wf)[2][2][2], const i32x4 (&xf)[4][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(xf[i][0]), "v"(xf[i][1]));
#pragma unroll
  for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(wf[i][0][0]), "v"(wf[i][0][1]), "v"(wf[i][1][0]), "v"(wf[i][1][1]));
}

__device__ __forceinline__ void wait_units_in_flight(int units) {  // 2 DMA instructions per unit per thread
  switch (units) {
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Timeline build (HWOCR_GEMM_ABLATE=10, tools/bench_gemm_timeline.py): wave 0 of every workgroup stamps the 100 MHz wall clock
// at the start of each tile's main loop, at its end and after the epilogue; results stay correct.
constexpr int TL_MAX = 256 * 64 * 4;
__device__ unsigned long long g_timeline[TL_MAX];

// STAGGER: the two wave-rows run half a phase apart (load segment | multiply segment, a barrier between segments): while
// waves 0-3 multiply, waves 4-7 fetch their fragments and vice versa, so the matrix pipe of every SIMD always has one of
// its two waves ready.  Costs a second barrier per phase; LDS hazards hold because every unit is overwritten 7 phases
// (14 segments) after... see the window derivation in DESIGN.md: last read of the old occupant at segment 2P-15, first
// DMA of the new one at 2P-14; the retiring wait of the late half moves from after its multiply to after its load segment.
template <int EPI, bool STAGGER, bool FP8, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_wide256_kernel(WideArgs a) {
  constexpr int ES = FP8 ? 1 : 2;  // bytes per operand element; a K tile is 128 bytes of every row either way
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int wr = w >> 2, wc = w & 3;
  const int ntiles = a.tilesM * a.tilesN;
  int tile = blockIdx.x;
  int m0, n0;
  auto origin_of = [&](int id) {  // XCD-contiguous chunks, then 4 row panels swept column-major (gemm_common.h)
    int tm, tn;
    tile_of_id(id, a.tilesM, a.tilesN, 4, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
  };

  // ---- staging plan.  Unit kind k, instruction j: this thread copies one 16-byte chunk of tile row ubase[k] + r8 + 128 j
  // (8 consecutive rows = 1 KiB per wave-instruction, lane-linear in LDS).
  const int r8 = lane >> 3, p = lane & 7;
  const int ubase[4] = {8 * w,                           // U0: activation rows 0..63 (+128)
                        (w >> 2) * 64 + (w & 3) * 8,     // U1: weight rows {0..31, 64..95} (+128)
                        (w >> 2) * 64 + (w & 3) * 8 + 32,  // U2: weight rows {32..63, 96..127} (+128)
                        64 + 8 * w};                     // U3: activation rows 64..127 (+128)
  const char* usrc[4][2];
  auto set_sources = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = ubase[k] + r8 + 128 * j;
        const int lc = p ^ ((row >> 1) & 7);
        usrc[k][j] = (k == 1 || k == 2) ? (const char*)a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw * ES + lc * 16
                                        : (const char*)a.X + (size_t)min(m0 + row, a.M - 1) * a.ldx * ES + lc * 16;
      }
  };
  const int nk = a.K * ES / 128;
  auto issue = [&](auto kind, int t) {  // unit (t, kind): 2 LDS-DMA instructions per thread
    constexpr int k = decltype(kind)::value;
    if (t < nk && ((ABL != 1 && ABL != 3 && ABL != 4) || t < 2)) {
      char* st = smem + (t & 1) * STAGE + ((k == 1 || k == 2) ? TILE : 0) + ubase[k] * 128;
      __builtin_amdgcn_global_load_lds((const void*)(usrc[k][0] + t * 128), LDS_PTR(st), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(usrc[k][1] + t * 128), LDS_PTR(st + 128 * 128), 16, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  const int total_units = 4 * nk;
  auto issue_prologue = [&]() {  // 7 units: K tile 0 and three quarters of K tile 1
    issue(K0{}, 0); issue(K1{}, 0); issue(K2{}, 0); issue(K3{}, 0);
    issue(K0{}, 1); issue(K1{}, 1); issue(K2{}, 1);
  };

  // fragment addresses inside a stage: row*128 + ((chunk ^ ((row>>1)&7)) << 4), rows = 16-aligned base + c
  const int sw = (c >> 1) & 7;
  const int xrow0 = (128 * wr + c) * 128;          // + (64 mh + 16 i) * 128
  const int wrow0 = TILE + (64 * wc + c) * 128;    // + (32 nh + 16 j) * 128
  i32x4 xf[4][2], wf[2][2][2];
  if constexpr (ABL == 4) {  // no LDS reads: the multiplies run on whatever the registers hold
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i][0] = xf[i][1] = i32x4{tid, i, tid, i};
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i][0][0] = wf[i][0][1] = wf[i][1][0] = wf[i][1][1] = i32x4{i, tid, i, tid};
  }
  auto read_x = [&](const char* st, int mh) {
    if constexpr (ABL == 4) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        xf[i][kk] = *(const i32x4*)(st + xrow0 + (64 * mh + 16 * i) * 128 + (((kk * 4 + q) ^ sw) << 4));
  };
  auto read_w = [&](const char* st, int nh, i32x4 (&dst)[2][2]) {
    if constexpr (ABL == 4) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        dst[j][kk] = *(const i32x4*)(st + wrow0 + (32 * nh + 16 * j) * 128 + (((kk * 4 + q) ^ sw) << 4));
  };
  auto phase_end = [&]() {
    if constexpr (ABL != 3) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- first tile: 7 units in flight, K tile 0 (units 0..3) landed
  origin_of(tile);
  set_sources();
  issue_prologue();
  wait_units_in_flight(max(0, min(3, total_units - 4)));
  phase_end();

  int tl_i = 0;
  auto stamp = [&](int what) {
    if constexpr (ABL == 10) {
      if (tid == 0) {
        const int slot = (blockIdx.x * 64 + tl_i) * 4 + what;
        if (slot < TL_MAX) {
          g_timeline[slot] = wall_clock64();
          if (what == 0) g_timeline[slot + 3] = clock64();  // shader clock beside the wall clock: the clock rate under load
        }
      }
    }
  };
  while (true) {
    stamp(0);
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (!STAGGER) {
      for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * STAGE;
        const int P = 4 * t;
        // ph0: (m0, n0)
        read_x(st, 0);
        read_w(st, 0, wf[0]);
        issue(K3{}, t + 1);  // unit P + 7
        if constexpr (ABL != 2) quadrant_mma<FP8, 0, 0, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        // ph1: (m0, n1)
        read_w(st, 1, wf[1]);
        issue(K0{}, t + 2);
        if constexpr (ABL != 2) quadrant_mma<FP8, 0, 1, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        // ph2: (m1, n1)
        read_x(st, 1);
        issue(K1{}, t + 2);
        if constexpr (ABL != 2) quadrant_mma<FP8, 1, 1, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        // ph3: (m1, n0); retire K tile t+1, keep the units issued behind it in flight
        issue(K2{}, t + 2);
        if constexpr (ABL != 2) quadrant_mma<FP8, 1, 0, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        wait_units_in_flight(max(0, min(3, total_units - 1 - (P + 7))));
        phase_end();
      }
    } else {
      const bool late = wr == 1;  // wave-uniform
      if (late) phase_end();
      for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * STAGE;
        const int keep = max(0, min(3, total_units - 1 - (4 * t + 7)));
        read_x(st, 0);
        read_w(st, 0, wf[0]);
        issue(K3{}, t + 1);
        phase_end();
        if constexpr (ABL != 2) quadrant_mma<FP8, 0, 0, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        read_w(st, 1, wf[1]);
        issue(K0{}, t + 2);
        phase_end();
        if constexpr (ABL != 2) quadrant_mma<FP8, 0, 1, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        read_x(st, 1);
        issue(K1{}, t + 2);
        phase_end();
        if constexpr (ABL != 2) quadrant_mma<FP8, 1, 1, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        phase_end();
        issue(K2{}, t + 2);
        if (late) wait_units_in_flight(keep);
        phase_end();
        if constexpr (ABL != 2) quadrant_mma<FP8, 1, 0, ABL == 11>(acc, wf, xf); else keep_alive(acc, wf, xf);
        if (!late) wait_units_in_flight(keep);
        phase_end();
      }
      if (!late) phase_end();
    }
    // Every wave is past its last fragment read and no DMA is in flight (the last K tile's waits kept 0 units).
    stamp(1);

    // ---- what the epilogue needs from global memory is fetched BEFORE the next tile's DMAs are issued: while a DMA is in
    // flight hipcc waits vmcnt(0) at the first use of an ordinary load's result, which would drain the prologue again
    const int em0 = m0, en0 = n0;  // origin of the finished tile
    bf16x4 bv[4];
    if constexpr (!is_glu<EPI>) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = en0 + 64 * wc + 16 * nt + 4 * q;
        bv[nt] = (a.bias && n < a.N) ? *(const bf16x4*)(a.bias + n) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      }
    }
    // residual rows of the finished tile, in the order the store passes consume them.  Two passes' worth (8 loads of a wave)
    // go out together ahead of the next tile's DMAs, the other two as soon as a pass has consumed its rows, so their latency
    // passes under the staging work (fetched pass by pass, each pass waited for its own four loads: ~7 us of a 49 us tile of
    // the K = 1280 projections)
    bf16x8 rs[2][4];
    auto load_res = [&](int pass) {
      const int n = en0 + 64 * wc + 8 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = em0 + 128 * wr + 32 * pass + 8 * i + (lane >> 3);
        rs[pass & 1][i] = *(const bf16x8*)(a.res + (size_t)min(m, a.M - 1) * a.ldres + min(n, a.N - 8));
      }
    };
    constexpr bool kResAhead = EPI == EPI_RESIDUAL && !FP8;  // (the fp8 form keeps 24 scale registers live here: it would spill)
    if constexpr (kResAhead) {
      load_res(0);
      load_res(1);
    }
    float xs[8];
    f32x4 ws[4];
    if constexpr (FP8) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) xs[mt] = a.xscale[min(em0 + 128 * wr + 16 * mt + c, a.M - 1)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) ws[nt] = *(const f32x4*)(a.wscale + min(en0 + 64 * wc + 16 * nt + 4 * q, a.N - 4));
    }

    // ---- next tile's prologue: its first K tiles land under this tile's epilogue
    tile += gridDim.x;
    const bool more = tile < ntiles;  // workgroup-uniform
    if (more) {
      origin_of(tile);
      set_sources();
      issue_prologue();
    }

    // ---- epilogue: lane (c,q) of tile (nt,mt) holds out[em0 + 128wr + 16mt + c][en0 + 64wc + 16nt + 4q .. +3]
    if constexpr (ABL == 5) {  // no epilogue: the accumulators only have to stay alive
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) sum += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
      if (sum == 12345.678f) a.out[0] = f2bf(sum);
    } else {
      if constexpr (FP8) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][mt][r] *= ws[nt][r] * xs[mt];
      }
      if constexpr (EPI == EPI_VIT_QKV) {
        // Vision QKV projection with the rotary embedding, the head split and the V transpose folded in (what
        // vit_rope_split_kernel does to a [tokens][3 DH] intermediate).  DH % 256 == 0: a tile is all q, all k or all v.
        // Every lane-derived index of this epilogue is recomputed per tile from a laundered copy of the lane id: left to itself
        // hipcc hoists them (two integer divisions among them) out of the tile loop and, the main loop having no register to
        // spare, carries them across it in scratch (2 VGPRs spilled in the bf16 instance, 23 in the E4M3 one).
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int lane = lane_l, c = lane & 15, q = lane >> 4;
        const hwocr_vit_split& vs = a.vs;
        const int hd = vs.hd, half = hd >> 1, quarter = hd >> 2, DH = vs.heads * hd;
        const int sec = en0 / DH, nsec = en0 - sec * DH;  // section (0 q, 1 k, 2 v) and the tile's first column inside it
        char* ep = smem + 2 * STAGE + w * 4096;
        if (sec < 2) {
          // q / k columns come as rotary pairs side by side (the weight rows were interleaved when they were bound), so a
          // lane's 4 consecutive columns of an MFMA tile are two whole pairs: columns 16nt + 4q + {0,1} = pair p (x1, x2),
          // {2,3} = pair p + 1, p = 8nt + 2q of the wave's 32 pairs; pair P of the section = head P / half, feature P % half.
          bf16* dst = (bf16*)(sec ? vs.K : vs.Q);
          const int P0 = (nsec + 64 * wc) >> 1;
          // feature index i of the lane's first pair in n-tile nt (the second is i + 1, same head, same axis) = (P0 + 8 nt + 2 q) % half:
          // one value kept, the other three stepped from it (8 <= half: a step wraps at most once) — four live registers fewer in
          // the rotation loop (the kernel sat at 256 VGPRs + 2 spilled)
          const int fi0 = (P0 + 2 * q) % half;
          // store side: chunk pch of a staged row = 8 features of group g = pch & 3, low (x1) half for pch < 4, high for pch >= 4
          const int pch = lane & 7;
          const int Pg = P0 + 8 * (pch & 3);
          const long dcol = (long)(Pg / half) * vs.tok_ld * hd + (Pg % half) + (pch >= 4 ? half : 0);
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
              const int mt = 2 * pass + mh, ml = 16 * mh + c;
              const int mc = min(em0 + 128 * wr + 16 * mt + c, a.M - 1);
              const int ph = vs.pos_h[mc], pw = vs.pos_w[mc];
              int i = fi0 - 8;
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) {
                i += 8;
                if (i >= half) i -= half;
                const int tab = (i < quarter ? ph * quarter + i : pw * quarter + i - quarter);
                const f32x2 cs = *(const f32x2*)(vs.cos_tab + tab), sn = *(const f32x2*)(vs.sin_tab + tab);
                float x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = rbf(acc[nt][mt][r] + bf2f(bv[nt][r]));
                bf16x2 lo, hi;  // fp32 products and sum, one rounding (HF apply_rotary_pos_emb_vision computes in float)
                lo[0] = f2bf(__fadd_rn(__fmul_rn(x[0], cs[0]), __fmul_rn(-x[1], sn[0])));
                hi[0] = f2bf(__fadd_rn(__fmul_rn(x[1], cs[0]), __fmul_rn(x[0], sn[0])));
                lo[1] = f2bf(__fadd_rn(__fmul_rn(x[2], cs[1]), __fmul_rn(-x[3], sn[1])));
                hi[1] = f2bf(__fadd_rn(__fmul_rn(x[3], cs[1]), __fmul_rn(x[2], sn[1])));
                *(bf16x2*)(ep + ml * 128 + ((nt ^ (ml & 7)) << 4) + q * 4) = lo;
                *(bf16x2*)(ep + ml * 128 + (((4 + nt) ^ (ml & 7)) << 4) + q * 4) = hi;
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            bf16x8 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 8 * i + (lane >> 3);
              v[i] = *(const bf16x8*)(ep + row * 128 + ((pch ^ (row & 7)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int m = em0 + 128 * wr + 32 * pass + 8 * i + (lane >> 3);
              if (m < a.M) *(bf16x8*)(dst + dcol + (long)m * hd) = v[i];
            }
          }
        } else {
          // v columns: transposed through LDS, 16 features x the wave's 128 rows per pass, so that every feature leaves as
          // 256 contiguous bytes of V^T.  Feature f of the pass sits at f * 256 B; its dword (row pair) j at j ^ (8 (f/4)).
          bf16* dst = (bf16*)vs.VT;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
              const int ml = 16 * mt + c;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(bf16*)(ep + (4 * q + r) * 256 + (((ml >> 1) ^ (q << 3)) << 2) + (ml & 1) * 2) = f2bf(acc[nt][mt][r] + bf2f(bv[nt][r]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            bf16x8 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int id = 64 * i + lane, f = id >> 4, ch = id & 15;
              v[i] = *(const bf16x8*)(ep + f * 256 + ((ch ^ (2 * ((f >> 2) & 3))) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int id = 64 * i + lane, f = id >> 4, ch = id & 15;
              const int n = nsec + 64 * wc + 16 * nt + f;       // feature of the v section
              const int row0 = em0 + 128 * wr + 8 * ch;
              if (row0 + 8 <= vs.tok_ld)
                *(bf16x8*)(dst + ((long)(n / hd) * hd + n % hd) * vs.tok_ld + row0) = v[i];
            }
          }
        }
      } else if constexpr (is_glu<EPI>) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const int m = em0 + 128 * wr + 16 * mt + c;
#pragma unroll
          for (int nt = 0; nt < 4; nt += 2) store_glu<EPI>(a, acc[nt][mt], acc[nt + 1][mt], m, en0 + 64 * wc + 16 * nt, q);
        }
      } else {
        // Through LDS so that HBM sees whole 128-byte rows: the fragment layout gives a lane 8 bytes of 16 different rows
        // per store (16 partial lines each); staged, a wave stores its 128 x 64 block as 16 instructions of 8 full rows.
        // Staging is private to the wave: 32 rows x 128 B at a time (4 passes), 16-byte chunk p of row r at p ^ (r & 7).
        char* ep = smem + 2 * STAGE + w * 4096;
        const int pch = lane & 7;
        const int n = en0 + 64 * wc + 8 * pch;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          if constexpr (EPI == EPI_RESIDUAL && !kResAhead) load_res(pass);
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const int mt = 2 * pass + mh;
              const int ml = 16 * mh + c;
              const bf16x4 o = epi_act4<EPI>(acc[nt][mt], bv[nt]);
              *(bf16x4*)(ep + ml * 128 + (((2 * nt + (q >> 1)) ^ (ml & 7)) << 4) + (q & 1) * 8) = o;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own wave's writes are in LDS before any lane reads them
          __builtin_amdgcn_wave_barrier();
          bf16x8 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3);
            v[i] = *(const bf16x8*)(ep + row * 128 + ((pch ^ (row & 7)) << 4));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and read back before the next pass overwrites them
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 32 * pass + 8 * i + (lane >> 3);
            const int m = em0 + 128 * wr + row;
            if (ABL == 6 ? (bf2f(v[i][0]) == 12345.678f) : (m < a.M && n < a.N)) {  // ABL 6: everything but the global stores
              if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = f2bf(bf2f(v[i][e]) + bf2f(rs[pass & 1][i][e]));
              }
              bf16x8* dst = (bf16x8*)(a.out + (size_t)m * a.ldo + n);
              if constexpr (ABL == 7) __builtin_nontemporal_store(v[i], dst);  // store policies: measured equal to the default one
              else if constexpr (ABL == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v[i]) : "memory");
              else if constexpr (ABL == 9) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v[i]) : "memory");
              else *dst = v[i];
            }
          }
          if constexpr (kResAhead)
            if (pass < 2) load_res(pass + 2);
        }
      }
    }
    stamp(2);
    ++tl_i;
    if (!more) break;
    // The next tile's K tile 0 must have landed.  In flight, oldest first: its 14 prologue DMAs, then this epilogue's stores.
    // vmcnt retires in issue order on gfx9-family parts (loads and stores share the one counter; the compiler's own waits rely
    // on it), so with the 16 stores of a full interior tile behind them "22 outstanding" = K tile 0 landed, 3 units + the stores
    // still flying — the store drain overlaps the first K tiles as it did when the workgroup simply ended.  Edge tiles skip
    // stores (fewer operations behind the DMAs): they wait for everything.
    constexpr bool kStores = ABL != 5 && ABL != 6;  // (ablation builds without the stores)
    if (kStores && em0 + BM <= a.M && en0 + BN <= a.N) {
      switch (max(0, min(3, total_units - 4))) {  // units of the prologue allowed to stay in flight (3 unless K < 128)
        case 3: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    } else {
      wait_units_in_flight(kStores ? 0 : max(0, min(3, total_units - 4)));
    }
    phase_end();
  }
}

// one workgroup per CU (its 160 KiB of LDS admit no second one), each walking tiles b, b + grid, ...
inline int persistent_grid(const WideArgs& b) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  const int ntiles = b.tilesM * b.tilesN, budget = hwocr_cu_budget();
  const int n = budget > 0 && budget < cus ? budget : cus;  // a CU-masked stream: one workgroup per CU it may use
  return ntiles < n ? ntiles : n;
}

template <int EPI, bool STAGGER, bool FP8>
void launch_one(const WideArgs& b, hipStream_t st) {
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_wide256_kernel<EPI, STAGGER, FP8>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)done;
  hipLaunchKernelGGL((gemm_wide256_kernel<EPI, STAGGER, FP8>), dim3(persistent_grid(b)), dim3(512), LDS_BYTES, st, b);
}
template <int EPI, bool FP8>
void launch(const WideArgs& a, hipStream_t st) {
  static const int variant = HWOCR_DIAG_ENV_INT("HWOCR_GEMM256", 3);  // 3: per-shape choice between the two forms (the product's rule)
  WideArgs b = a;
  b.tilesM = (a.M + BM - 1) / BM;
  b.tilesN = (a.N + BN - 1) / BN;
  // the four-wave form (gemm256w4.hip) where it exists - bf16, not the fused vision QKV, at least two K tiles - and is the faster one
  // (its own rule; HWOCR_GEMM256=4: wherever it exists, =2 / =1: never)
  if constexpr (!FP8 && EPI != EPI_VIT_QKV) {
    if (variant >= 3 && hwocr_gemm_wide256_w4(b, EPI, variant == 4, st)) return;
  }
  if (hwocr_plan_on()) {
    hwocr_plan_note("gemm_wide256_kernel<epi=%d,%s,%s> M=%d N=%d K=%d tiles=%d grid=%d rounds=%d ktiles=%d", EPI,
                    variant == 1 ? "lockstep" : "stagger", FP8 ? "e4m3" : "bf16", a.M, a.N, a.K, b.tilesM * b.tilesN, persistent_grid(b),
                    (b.tilesM * b.tilesN + persistent_grid(b) - 1) / persistent_grid(b), a.K * (FP8 ? 1 : 2) / 128);
    return;
  }
#ifdef HWOCR_DIAG  // diagnostic builds only (build.use_diag_library(), csrc/diag/): kernel variants that give WRONG results by construction
  if constexpr (EPI == EPI_LINEAR && !FP8) {
    static const int ablate = [] { const char* e = getenv("HWOCR_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    if (ablate >= 7 && ablate <= 9) {  // store policies of the epilogue (results stay correct): 7 nt, 8 sc1, 9 sc0 sc1
      auto k = ablate == 7 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 7>
               : ablate == 8 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 8>
                             : gemm_wide256_kernel<EPI_LINEAR, true, false, 9>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      hipLaunchKernelGGL(k, dim3(persistent_grid(b)), dim3(512), LDS_BYTES, st, b);
      return;
    }
    if (ablate == 11) {
      auto k = gemm_wide256_kernel<EPI_LINEAR, true, false, 11>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      hipLaunchKernelGGL(k, dim3(persistent_grid(b)), dim3(512), LDS_BYTES, st, b);
      return;
    }
    if (ablate == 10) {
      auto k = gemm_wide256_kernel<EPI_LINEAR, true, false, 10>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      hipLaunchKernelGGL(k, dim3(persistent_grid(b)), dim3(512), LDS_BYTES, st, b);
      return;
    }
    if (ablate >= 1 && ablate <= 6) {
      auto k = ablate == 1 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 1>
               : ablate == 2 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 2>
               : ablate == 3 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 3>
               : ablate == 4 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 4>
               : ablate == 5 ? gemm_wide256_kernel<EPI_LINEAR, true, false, 5>
                             : gemm_wide256_kernel<EPI_LINEAR, true, false, 6>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      hipLaunchKernelGGL(k, dim3(persistent_grid(b)), dim3(512), LDS_BYTES, st, b);
      return;
    }
  }
#endif
  if (variant == 1) launch_one<EPI, false, FP8>(b, st);
  else launch_one<EPI, true, FP8>(b, st);
}
template <bool FP8>
int dispatch(const WideArgs& a, int epi, hipStream_t stream) {
  switch (epi) {
    case EPI_LINEAR: launch<EPI_LINEAR, FP8>(a, stream); break;
    case EPI_RESIDUAL: launch<EPI_RESIDUAL, FP8>(a, stream); break;
    case EPI_QUICKGELU: launch<EPI_QUICKGELU, FP8>(a, stream); break;
    case EPI_GELU: launch<EPI_GELU, FP8>(a, stream); break;
    case EPI_SWIGLU: launch<EPI_SWIGLU, FP8>(a, stream); break;
    case EPI_GELU_TANH: launch<EPI_GELU_TANH, FP8>(a, stream); break;
    case EPI_GEGLU: launch<EPI_GEGLU, FP8>(a, stream); break;
    case EPI_VIT_QKV: launch<EPI_VIT_QKV, FP8>(a, stream); break;
    default: return HWOCR_EINVAL;
  }
  return hwocr_plan_on() ? HWOCR_OK : hwocr_launch_status();
}

}  // namespace

int hwocr_gemm_wide256(const WideArgs& a, int epi, hipStream_t stream) { return dispatch<false>(a, epi, stream); }

#ifdef HWOCR_DIAG
// timeline stamps of the last HWOCR_GEMM_ABLATE=10 launch: [workgroup][tile slot 0..63][start, loop end, epilogue end, -] ticks of 10 ns
extern "C" int hwocr_debug_gemm_timeline(unsigned long long* host, int n) {
  if (!host || n <= 0 || n > TL_MAX) return HWOCR_EINVAL;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? HWOCR_OK : HWOCR_ELAUNCH;
}
#endif

// X and W hold E4M3 bytes (ldx / ldw / K in elements = bytes), a.xscale / a.wscale their per-row fp32 scales
int hwocr_gemm_wide256_fp8(const WideArgs& a, int epi, hipStream_t stream) {
  if (!a.xscale || !a.wscale || (a.K % 128) || (a.ldx % 16) || (a.ldw % 16) || a.N < 4) return HWOCR_EINVAL;
  return dispatch<true>(a, epi, stream);
}

// QKV projection of a vision block with rotary + head split + V transpose in the epilogue (entry point: gemm.hip)
int hwocr_gemm_wide256_vit_qkv(const WideArgs& a, bool fp8, hipStream_t stream) {
  return fp8 ? dispatch<true>(a, EPI_VIT_QKV, stream) : dispatch<false>(a, EPI_VIT_QKV, stream);
}
