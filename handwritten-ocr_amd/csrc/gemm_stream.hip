// Decode-step GEMM: out[Bsz <= 256][N] = x[Bsz][K] . W[N][K]^T with W streamed from HBM exactly once
// (gemm_stream_kernel: up to 128 rows; gemm_stream256_kernel below: 129..256, the default 252 reads in flight).
//
// A decode step multiplies the same activation rows (one per read in flight) by every decoder weight.  The
// first decode kernel (gemm_skinny_kernel: weights HBM -> VGPR in two register stages, x chunks by LDS-DMA, 2-4 waves)
// kept 4-8 KB in flight per wave and measured 1.7-3.1 TB/s of weights at 126 rows.  This kernel:
//   * W is the fragment-tiled copy ([N/16][K/32][lane][8], hwocr_tile_weights): the K axis of a 16-row tile is ONE
//     contiguous stream and a 1 KiB LDS-DMA instruction delivers one MFMA A fragment in lane order — the LDS image
//     needs no swizzle, fragment reads are lane-linear ds_read_b128.
//   * x K-tiles ride the same ring (LDS-DMA from L2): rows of 128 B, one instruction = 8 whole rows (whole cache
//     lines), 16-byte chunks XOR-swizzled on the SOURCE address so that the B fragment reads are conflict-free.
//   * ring of 3 stages of one 64-wide K tile: while tile t is multiplied, t+1 and t+2 are in flight; one raw s_barrier
//     per K tile; each wave waits with a COUNTED vmcnt for its own DMAs only.
//   * one workgroup of 16 waves per CU.  Staging: wave w brings in weight tile w of the group and a share of the x rows.
//     Multiply: 16 x 1 (wave w = tile w x all row tiles) up to 64 rows, 2 row halves x 8 tile pairs at 65..128 rows.
//     Weight tiles are dealt to workgroups as balanced contiguous ranges of at most 16 tiles, so grid.x * splitk can be
//     made ~256 (one round) whatever N is: 37888 rows (Qwen2.5-VL-7B gate/up) = 1184 tile pairs = 4.6 pairs per CU, 92 %
//     balance.  Gated epilogues in the 16 x 1 form: the gate tile sits in the even wave and the up tile in the odd one;
//     the up accumulators cross through the idle ring once at the end.
// What bounds it (7B gate/up, 271 MB of weights, 126 rows): every workgroup must also pull the whole activation tile
// (126 x K x 2 B) through L2 -> LDS, and the bytes entering the CUs top out near 8 TB/s chip-wide = 31-33 GB/s per CU —
// the same per-CU rate the 256x256 prefill GEMM runs at, and the same whether the bytes come by LDS-DMA or by
// global_load + ds_write.  Measured on the way: 4 waves 2.9 TB/s of weights, 8 waves 4.0, 16 waves 4.4-4.5 (16 rows
// instead of 126, i.e. an 8x smaller x tile: 5.6 TB/s); time = (W + 256 x-tiles) / 8 TB/s within 5 % on every shape.
// A variant with the weights HBM -> VGPR in a 4-deep hand-unrolled register ring (exact compiler vmcnt waits once the
// loop has no separate prologue and no load under a t-dependent branch) and x by plain loads + ds_write moved bytes into
// the CUs at the same rate but re-read weight tiles for idle waves: 3.3 TB/s; dropped.
// Later finding (252 rows): the cost is per loop trip and per DMA instruction rather than per byte - the 32-wide-K kernel's
// skeleton alone (4-byte DMAs, no fragment reads, no MFMA) takes 31 of the 41 us of the 2B gate/up GEMM, about 33 1-KiB DMA
// instructions per us per CU - hence the 64-wide-K form with 16 row tiles below (gemm_stream_kernel<16, EPI, 2, WT>).
// Round 2 re-measured the two remaining suspects on the two-row-block form at 252 rows (tools/bench_decode_plan.py, weights cold):
// smaller stages = a deeper ring (8 / 10 / 12 weight slots instead of 16: 5 / 4 / 4 stages instead of 3) and non-temporal weight
// DMAs.  2B gate/up 26.5 us with the 3-stage ring against 27.5-28.6 with the deeper ones, 26.5 against 26.5-28.6 with nt; qkv / o /
// down likewise within +-1 us.  Bytes in flight are not what holds the kernel at ~31 GB/s per CU; the guide's own LDS-DMA ring
// GEMM (MI355X_MICROARCH.md, rows ring-gemm / ring-vs-splitk: 25 MB of weights x 192 rows in 14.7-19.7 us = 1.3-1.7 TB/s of
// weights) is slower per weight byte than this kernel (55 MB x 252 rows in 26.5 us = 2.1 TB/s).
// Epilogues as gemm_skinny: PARTIAL fp32 split-K slabs, LINEAR (+bias), SWIGLU / GEGLU on interleaved gate/up tile pairs.
#include "gemm_common.h"
#include <cstdlib>

using namespace gemm;

namespace {

constexpr int WG_TILES = 16;               // weight tiles per workgroup
constexpr int WAVES = 16;
constexpr int WBYTES = WG_TILES * 2048;    // one K tile (64) of 16 weight tiles' fragments
// stages of the ring of gemm_stream_kernel<MT, ., ., WT>: as many 64-wide K tiles (x tile + WT weight-tile slots) as fit in
// 160 KiB, at most 6 (wait_dma counts up to 15 outstanding instructions: 3 per K tile and wave)
constexpr int stream_stages(int MT, int WT, bool W8 = false) {
  const int stage = 2 * MT * 1024 + WT * (W8 ? 1024 : 2048);
  const int n = (160 * 1024) / stage;
  return n > 6 ? 6 : n;
}

__device__ __forceinline__ void wait_dma(int pending) {  // leave `pending` of this wave's DMA instructions in flight
  switch (pending) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // more than 20 never occurs (<= 5 tiles x 4 instructions)
  }
}

// 8 E4M3 codes (two dwords) -> the bf16 fragment they stand for: every E4M3 value is exactly representable in bf16, so the
// decode GEMMs of the fp8 configuration multiply exactly the stored codes (the per-feature scale goes into the epilogue).
__device__ __forceinline__ bf16x8 e4m3x8_to_bf16(int lo, int hi) {
  // v_cvt_scalef32_pk_bf16_fp8 (gfx950): two codes -> two packed bf16 per instruction, scale 1 (4 instead of 8 conversions per
  // fragment: the conversions are VALU work on the K-tile critical path of a kernel that is not weight-byte bound)
  const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  return __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3), __builtin_shufflevector(c, d, 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
}

// MT: 16-row activation tiles (Bsz <= 16 MT).  16 waves = MSPLIT (row halves) x 16/MSPLIT (tile groups): a wave multiplies
// MT/MSPLIT row tiles by MSPLIT weight tiles.  MSPLIT = 2 halves the x fragments every wave has to read from LDS (all 16
// waves reading the whole x tile is 288 KB of ds_read per K tile at 128 rows - as long as the DMA of that K tile takes).
// Staging is independent of that split: wave w brings in weight tile w of the group and x rows {8(w + 16e)}.
// grid = (tile groups, K slices).
// WT: weight tiles a workgroup can own (its LDS slots).  The 16-row-tile form (129..256 rows) exists only with WT = 8 or 10:
// its x tile alone is 32 KiB per K tile, and the weight slots a workgroup does not use (it usually owns 2-5 tiles: N / 16
// tiles over 256 / splitk groups) are what pays for a third and fourth stage.  Why it exists beside gemm_stream256_kernel:
// that kernel's loop SKELETON (counted wait + 16-wave barrier + DMA issue per 32-wide K tile) measures 31 of the 41 us of the 2B
// gate/up GEMM with every byte and every MFMA removed; 64-wide K tiles halve the number of trips.
// W8: the weights are E4M3 codes in the byte-tiled layout of hwocr_tile_weights_fp8 ([N/16][K/64][lane][16 B] = a lane's 8
// codes of the first 32-wide half of the K tile, then of the second): ONE 1-KiB DMA per weight tile and K tile (half the bytes
// entering the CU for the weights), converted to bf16 fragments in registers; a.wscale[n] scales output feature n.
template <int MT, int EPI, int MSPLIT, int WT = WG_TILES, bool W8 = false>
__global__ __launch_bounds__(64 * WAVES) void gemm_stream_kernel(StreamArgs a) {
  constexpr int WSLOT = W8 ? 1024 : 2048;           // bytes of one weight tile's K tile
  constexpr int WBYTES = WT * WSLOT;                // (shadows the 16-tile constant)
  constexpr int NWN = WAVES / MSPLIT;               // waves along N
  constexpr int NTW = WG_TILES / NWN;               // weight tiles per wave (= MSPLIT)
  constexpr int MTW = MT / MSPLIT;                  // row tiles per wave
  constexpr int XFR = 2 * MT;                       // x DMA instructions (8 rows x 128 B = 1 KiB each) per K tile
  constexpr int XPW = (XFR + WAVES - 1) / WAVES;    // staged per wave
  constexpr int XBYTES = XFR * 1024;
  constexpr int STAGE = XBYTES + WBYTES;
  constexpr int NSTAGE = stream_stages(MT, WT, W8);
  static_assert(NSTAGE * STAGE <= 160 * 1024, "ring does not fit in LDS");
  constexpr int DIST = NSTAGE - 1;                  // K tiles in flight ahead of the one being multiplied
  static_assert(MT % MSPLIT == 0 && (MSPLIT == 1 || MSPLIT == 2), "row tiles split evenly over the wave rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int wn = w % NWN, wm = w / NWN;

  // ---- this workgroup's weight tiles [t0, t1) (gated epilogues: whole gate/up pairs) and K tiles [kt0, kt0 + nk)
  constexpr int UNIT = is_glu<EPI> ? 2 : 1;
  const int units = (a.N >> 4) / UNIT;
  const int t0 = (int)((long)blockIdx.x * units / gridDim.x) * UNIT;
  const int t1 = (int)((long)(blockIdx.x + 1) * units / gridDim.x) * UNIT;
  const int my0 = t0 + wn * NTW;                    // first tile this wave multiplies
  const int mine = max(0, min(NTW, t1 - my0));      // wave-uniform
  const bool stage_w = t0 + w < t1;                 // this wave stages tile t0 + w
  const int kt0 = blockIdx.y * a.ktiles_per_slice;
  const int nk = min(a.ktiles_per_slice, (a.K >> 6) - kt0);
  const int r0 = blockIdx.z * (16 * MT);            // row block (grid.z = 2: 129..256 rows as two blocks of 128, see launcher)

  // ---- DMA sources.  x instruction f stages rows 8f .. 8f+7: lane l <- 16-byte chunk (l&7) ^ (l>>3) of row 8f + (l>>3),
  // so chunk ch of row r sits at LDS position ch ^ (r & 7) of its 128-byte row
  const bf16* xsrc[XPW];
#pragma unroll
  for (int e = 0; e < XPW; ++e) {
    const int f = w + WAVES * e;
    xsrc[e] = a.X + (size_t)min(r0 + 8 * f + (lane >> 3), a.Bsz - 1) * a.ldx + (size_t)kt0 * 64 + 8 * ((lane & 7) ^ (lane >> 3));
  }
  // bf16: + t*1024 + h*512 elements; E4M3: a K tile of a weight tile is 1024 bytes = 512 "elements" of this pointer type
  const bf16* wsrc = W8 ? a.W + ((size_t)(t0 + w) * (a.K >> 6) + (size_t)kt0) * 512 + lane * 8
                        : a.W + ((size_t)(t0 + w) * (a.K >> 5) + (size_t)kt0 * 2) * 512 + lane * 8;
  const int n_dma = (XFR > w ? (XFR - w + WAVES - 1) / WAVES : 0) + (stage_w ? (W8 ? 1 : 2) : 0);  // DMA instructions per stage

  auto issue = [&](int t) {
    char* st = smem + (t % NSTAGE) * STAGE;
#pragma unroll
    for (int e = 0; e < XPW; ++e)
      if (w + WAVES * e < XFR)
        __builtin_amdgcn_global_load_lds((const void*)(xsrc[e] + t * 64), LDS_PTR(st + (w + WAVES * e) * 1024), 16, 0, 0);
    if (stage_w) {
      if constexpr (W8) {
        __builtin_amdgcn_global_load_lds((const void*)(wsrc + (size_t)t * 512), LDS_PTR(st + XBYTES + w * 1024), 16, 0, 0);
      } else {
        const bf16* s = wsrc + (size_t)t * 1024;
        char* d = st + XBYTES + w * 2048;
        __builtin_amdgcn_global_load_lds((const void*)s, LDS_PTR(d), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(s + 512), LDS_PTR(d + 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[NTW][MTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nk) issue(d);
  for (int t = 0; t < nk; ++t) {
    wait_dma(min(DIST - 1, nk - 1 - t) * n_dma);  // K tile t of this wave has landed; later ones may still fly
    __builtin_amdgcn_s_barrier();                  // ... of every wave; and every wave is done reading K tile t-1
    __builtin_amdgcn_sched_barrier(0);
    if (t + DIST < nk) issue(t + DIST);            // into the slot K tile t-1 occupied
    const char* st = smem + (t % NSTAGE) * STAGE;
    if (mine == 0) continue;  // wave columns without a tile read no fragments (16 KiB per wave and K tile at 16 row tiles)
    if constexpr (MT == 16) {
      // 4 waves per SIMD = 128 registers: 64 accumulators leave room for ONE 32-wide half of the fragments at a time
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 xh[MTW], wh[NTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i)
          xh[i] = *(const bf16x8*)(st + (16 * (wm * MTW + i) + c) * 128 + (((4 * h + q) ^ (c & 7)) << 4));
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if constexpr (W8) {
            const int2 raw = *(const int2*)(st + XBYTES + (wn * NTW + j) * 1024 + lane * 16 + h * 8);
            wh[j] = e4m3x8_to_bf16(raw.x, raw.y);
          } else {
            wh[j] = *(const bf16x8*)(st + XBYTES + (wn * NTW + j) * 2048 + h * 1024 + lane * 16);
          }
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          if (j < mine) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh[i], acc[j][i], 0, 0, 0);
          }
      }
      continue;
    }
    bf16x8 xf[MTW][2], wf[NTW][2];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        xf[i][h] = *(const bf16x8*)(st + (16 * (wm * MTW + i) + c) * 128 + (((4 * h + q) ^ (c & 7)) << 4));
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if constexpr (W8) {
        const int4 raw = *(const int4*)(st + XBYTES + (wn * NTW + j) * 1024 + lane * 16);
        wf[j][0] = e4m3x8_to_bf16(raw.x, raw.y);
        wf[j][1] = e4m3x8_to_bf16(raw.z, raw.w);
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) wf[j][h] = *(const bf16x8*)(st + XBYTES + (wn * NTW + j) * 2048 + h * 1024 + lane * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      if (j < mine) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < MTW; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][h], xf[i][h], acc[j][i], 0, 0, 0);
      }
  }

  if constexpr (W8) {  // per-output-feature scale of the E4M3 codes
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      if (j < mine) {
        const f32x4 sc = *(const f32x4*)(a.wscale + 16 * (my0 + j) + 4 * q);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][i][r] *= sc[r];
      }
  }
  // ---- epilogue: lane (c,q): acc[j][i][r] = out[16 (wm MTW + i) + c][16 (my0 + j) + 4 q + r]
  if constexpr (is_glu<EPI> && NTW == 1) {
    // gate tile in the even wave, up tile in the odd one: the up accumulators cross through the (now idle) ring
    __builtin_amdgcn_s_barrier();  // every wave is done reading the last K tile
    float* xch = (float*)smem + (w >> 1) * (MTW * 256);
    if (w & 1) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) *(f32x4*)(xch + (i * 64 + lane) * 4) = acc[0][i];
    }
    __builtin_amdgcn_s_barrier();
    if (!(w & 1) && mine > 0) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int m = r0 + 16 * i + c;
        const f32x4 up = *(const f32x4*)(xch + (i * 64 + lane) * 4);
        if (m < a.Bsz) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(glu_gate<EPI>(rbf(acc[0][i][r]))) * rbf(up[r]));
          *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + 8 * my0 + 4 * q) = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int m = r0 + 16 * (wm * MTW + i) + c;
    if (m >= a.Bsz) continue;
    if constexpr (is_glu<EPI>) {
#pragma unroll
      for (int j = 0; j + 1 < NTW; j += 2)
        if (j < mine) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(glu_gate<EPI>(rbf(acc[j][i][r]))) * rbf(acc[j + 1][i][r]));
          *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + 8 * (my0 + j) + 4 * q) = o;
        }
    } else {
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (j < mine) {
          const int n = 16 * (my0 + j) + 4 * q;
          if constexpr (EPI == EPI_PARTIAL) {
            *(f32x4*)((float*)a.out + ((size_t)blockIdx.y * a.Bsz + m) * a.ldo + n) = acc[j][i];
          } else {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r];
            if (a.bias) {
              const bf16x4 bb = *(const bf16x4*)(a.bias + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(bb[r]);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
            *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) = o;
          }
        }
    }
  }
}

// 129..256 rows: the x tile of a 64-wide K tile alone is 32 KiB and only two stages would fit, i.e. nothing in flight
// while a stage is awaited (measured: 4.2 us per K tile).  This variant walks K in 32-wide tiles instead: a stage is 16
// x fragments + 16 weight fragments of 1 KiB (one DMA instruction of each per wave, both already in MFMA operand order:
// lane (c, q) of x fragment i <- x[16 i + c][32 t + 8 q ..]), four stages, three K tiles in flight.  Waves: 2 row halves
// x 8 tile pairs, as MSPLIT = 2 above.
template <int EPI>
__global__ __launch_bounds__(64 * WAVES) void gemm_stream256_kernel(StreamArgs a) {
  constexpr int MT = 16, MTW = 8, NTW = 2, NWN = 8;
  constexpr int STAGE = (MT + WG_TILES) * 1024, NSTAGE = 4, DIST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int wn = w % NWN, wm = w / NWN;
  constexpr int UNIT = is_glu<EPI> ? 2 : 1;
  const int units = (a.N >> 4) / UNIT;
  const int t0 = (int)((long)blockIdx.x * units / gridDim.x) * UNIT;
  const int t1 = (int)((long)(blockIdx.x + 1) * units / gridDim.x) * UNIT;
  const int my0 = t0 + wn * NTW;
  const int mine = max(0, min(NTW, t1 - my0));
  const bool stage_w = t0 + w < t1;
  const int kt0 = blockIdx.y * a.ktiles_per_slice * 2;                       // in 32-wide tiles
  const int nk = min(a.ktiles_per_slice * 2, (a.K >> 5) - kt0);
  const bf16* xsrc = a.X + (size_t)min(16 * w + c, a.Bsz - 1) * a.ldx + (size_t)kt0 * 32 + 8 * q;  // x fragment w
  const bf16* wsrc = a.W + ((size_t)(t0 + w) * (a.K >> 5) + kt0) * 512 + lane * 8;
  const int n_dma = 1 + (stage_w ? 1 : 0);
  auto issue = [&](int t) {
    char* st = smem + (t % NSTAGE) * STAGE;
    __builtin_amdgcn_global_load_lds((const void*)(xsrc + t * 32), LDS_PTR(st + w * 1024), 16, 0, 0);
    if (stage_w) __builtin_amdgcn_global_load_lds((const void*)(wsrc + (size_t)t * 512), LDS_PTR(st + (MT + w) * 1024), 16, 0, 0);
  };
  f32x4 acc[NTW][MTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nk) issue(d);
  for (int t = 0; t < nk; ++t) {
    wait_dma(min(DIST - 1, nk - 1 - t) * n_dma);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + DIST < nk) issue(t + DIST);
    const char* st = smem + (t % NSTAGE) * STAGE;
    bf16x8 xf[MTW], wf[NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) xf[i] = *(const bf16x8*)(st + (wm * MTW + i) * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < NTW; ++j) wf[j] = *(const bf16x8*)(st + (MT + wn * NTW + j) * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      if (j < mine) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int m = 16 * (wm * MTW + i) + c;
    if (m >= a.Bsz) continue;
    if constexpr (is_glu<EPI>) {
      if (mine > 0) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(glu_gate<EPI>(rbf(acc[0][i][r]))) * rbf(acc[1][i][r]));
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + 8 * my0 + 4 * q) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (j < mine) {
          const int n = 16 * (my0 + j) + 4 * q;
          if constexpr (EPI == EPI_PARTIAL) {
            *(f32x4*)((float*)a.out + ((size_t)blockIdx.y * a.Bsz + m) * a.ldo + n) = acc[j][i];
          } else {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r];
            if (a.bias) {
              const bf16x4 bb = *(const bf16x4*)(a.bias + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(bb[r]);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
            *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) = o;
          }
        }
    }
  }
}

template <int EPI>
void launch256(const StreamArgs& a, dim3 grid, hipStream_t st) {
  constexpr int LDS = 4 * (16 + WG_TILES) * 1024;
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_stream256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    return true;
  }();
  (void)done;
  hipLaunchKernelGGL((gemm_stream256_kernel<EPI>), grid, dim3(64 * WAVES), LDS, st, a);
}

template <int MT, int EPI, int MSPLIT, int WT, bool W8>
void launch_one(const StreamArgs& a, dim3 grid, hipStream_t st) {
  constexpr int STAGE = 2 * MT * 1024 + WT * (W8 ? 1024 : 2048);
  constexpr int LDS = stream_stages(MT, WT, W8) * STAGE;
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<MT, EPI, MSPLIT, WT, W8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS);
    return true;
  }();
  (void)done;
  hipLaunchKernelGGL((gemm_stream_kernel<MT, EPI, MSPLIT, WT, W8>), grid, dim3(64 * WAVES), LDS, st, a);
}

template <int MT, int MSPLIT, int WT, bool W8>
int launch_epi(const StreamArgs& a, int epi, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_LINEAR: launch_one<MT, EPI_LINEAR, MSPLIT, WT, W8>(a, grid, st); break;
    case EPI_SWIGLU: launch_one<MT, EPI_SWIGLU, MSPLIT, WT, W8>(a, grid, st); break;
    case EPI_GEGLU: launch_one<MT, EPI_GEGLU, MSPLIT, WT, W8>(a, grid, st); break;
    case EPI_PARTIAL: launch_one<MT, EPI_PARTIAL, MSPLIT, WT, W8>(a, grid, st); break;
    default: return HWOCR_EINVAL;
  }
  return hwocr_launch_status();
}
template <int MT, int MSPLIT, int WT = WG_TILES>
int launch_mt(const StreamArgs& a, int epi, dim3 grid, hipStream_t st) {
  return a.wscale ? launch_epi<MT, MSPLIT, WT, true>(a, epi, grid, st) : launch_epi<MT, MSPLIT, WT, false>(a, epi, grid, st);
}

}  // namespace

// Shapes this kernel takes: fragment-tiled W, Bsz <= 256, K % 64 == 0, every K slice non-empty.
// The choice of kernel instance and grid is made by plan_stream alone, so that hwocr_gemm_stream_variant (what the parity
// tests ask: "which instance would this call run?") and the launcher cannot disagree.
namespace {
enum StreamKind : int { SK_MT1, SK_MT2, SK_MT4, SK_MT8_M1, SK_MT8_M2, SK_MT8_M2_R2, SK_K64_W8, SK_K64_W10, SK_K32_256, SK_K64_W16 };
const char* const kStreamKindName[] = {
    "gemm_stream_kernel<1,1,16>",        "gemm_stream_kernel<2,1,16>",  "gemm_stream_kernel<4,1,16>",
    "gemm_stream_kernel<8,1,16>",        "gemm_stream_kernel<8,2,16>",  "gemm_stream_kernel<8,2,16>/rowblocks2",
    "gemm_stream_kernel<16,2,8>",        "gemm_stream_kernel<16,2,10>", "gemm_stream256_kernel", "gemm_stream_kernel<16,2,16>"};
struct StreamPlan { int kind; dim3 grid; int ktiles_per_slice; };

// w8: E4M3 weights (a weight tile's K tile is 1 KiB instead of 2): the 16-slot form fits three stages at 129..256 rows too, so
// the 32-wide-K kernel is never needed
int plan_stream(int Bsz, int N, int K, int epi, int splitk, bool w8, StreamPlan& p) {
  if (Bsz < 1 || Bsz > 256 || (K % 64) || (N % 16) || splitk < 1) return HWOCR_EINVAL;
  if (epi != EPI_LINEAR && epi != EPI_SWIGLU && epi != EPI_GEGLU && epi != EPI_PARTIAL) return HWOCR_EINVAL;
  const int ktiles = K / 64;
  p.ktiles_per_slice = (ktiles + splitk - 1) / splitk;
  if ((splitk - 1) * p.ktiles_per_slice >= ktiles) return HWOCR_EINVAL;  // an empty slice would leave its slab unwritten
  const int unit = (epi == EPI_SWIGLU || epi == EPI_GEGLU) ? 2 : 1;
  const int units = N / 16 / unit, per_wg = WG_TILES / unit;
  // one workgroup per CU: as many tile groups as fill the chip once with this split (more only if a group would exceed
  // 16 tiles; then whole rounds of 256 workgroups)
  int groups = (units + per_wg - 1) / per_wg;
  const int want = 256 / splitk;
  if (groups < want) groups = want;
  else if (groups * splitk > 256) groups = ((groups * splitk + 255) / 256 * 256) / splitk;
  if (groups > units) groups = units;
  // the balanced split must not hand any group more than per_wg units
  while ((units + groups - 1) / groups > per_wg) ++groups;
  p.grid = dim3(groups, splitk);
  const int mt = (Bsz + 15) / 16;
  static const int split8 = HWOCR_DIAG_ENV_INT("HWOCR_STREAM_MSPLIT", 2);
  if (mt <= 1) { p.kind = SK_MT1; return HWOCR_OK; }
  if (mt <= 2) { p.kind = SK_MT2; return HWOCR_OK; }
  if (mt <= 4) { p.kind = SK_MT4; return HWOCR_OK; }
  if (mt <= 8) { p.kind = split8 == 2 ? SK_MT8_M2 : SK_MT8_M1; return HWOCR_OK; }
  // 129..256 rows.  A workgroup that owns at most 10 weight tiles (every decoder GEMM of the 2B / 3B / 7B shapes; not the LM
  // head) takes the 64-wide-K form; HWOCR_STREAM_K64=0: always the 32-wide-K kernel
  static const bool k64 = HWOCR_DIAG_ENV_INT("HWOCR_STREAM_K64", 1) != 0;
  const int tiles_per_wg = ((units + groups - 1) / groups) * unit;
  // What a workgroup pulls in per K tile is cache lines: 16 rows-tiles x 16 lines of x + 16 lines per weight tile.  Below 8
  // tiles per workgroup two row blocks of 128 (grid.z = 2, half as many tile groups, so twice the tiles each) need fewer:
  // 128 + 32 t against 256 + 16 t lines (2B gate/up: 268 vs 326).  HWOCR_STREAM_R2=0 disables.
  static const bool r2 = HWOCR_DIAG_ENV_INT("HWOCR_STREAM_R2", 1) != 0;
  if (k64 && r2 && tiles_per_wg < 8) {
    int g2 = 256 / (2 * splitk);  // never more than one round of workgroups (a few left over for a second round cost a whole trip count)
    if (g2 < 1) g2 = 1;
    if (g2 > units) g2 = units;
    while ((units + g2 - 1) / g2 > per_wg) ++g2;
    p.kind = SK_MT8_M2_R2;
    p.grid = dim3(g2, splitk, 2);
    return HWOCR_OK;
  }
  if (k64 && tiles_per_wg <= 8) { p.kind = SK_K64_W8; return HWOCR_OK; }
  if (k64 && tiles_per_wg <= 10) { p.kind = SK_K64_W10; return HWOCR_OK; }  // 52 KiB stages, 3 of them (7B gate/up: 5 pairs)
  p.kind = w8 ? SK_K64_W16 : SK_K32_256;
  return HWOCR_OK;
}
}  // namespace

int hwocr_gemm_stream_variant(int Bsz, int N, int K, int epi, int splitk, bool w8, const char** name) {
  StreamPlan p;
  const int rc = plan_stream(Bsz, N, K, epi, splitk, w8, p);
  if (rc == HWOCR_OK && name) *name = kStreamKindName[p.kind];
  return rc;
}

int hwocr_gemm_stream(StreamArgs a, int epi, int splitk, hipStream_t stream) {
  StreamPlan p;
  const int rc = plan_stream(a.Bsz, a.N, a.K, epi, splitk, a.wscale != nullptr, p);
  if (rc != HWOCR_OK) return rc;
  a.ktiles_per_slice = p.ktiles_per_slice;
  switch (p.kind) {
    case SK_MT1: return launch_mt<1, 1>(a, epi, p.grid, stream);
    case SK_MT2: return launch_mt<2, 1>(a, epi, p.grid, stream);
    case SK_MT4: return launch_mt<4, 1>(a, epi, p.grid, stream);
    case SK_MT8_M1: return launch_mt<8, 1>(a, epi, p.grid, stream);
    case SK_MT8_M2:
    case SK_MT8_M2_R2: return launch_mt<8, 2>(a, epi, p.grid, stream);
    case SK_K64_W8: return launch_mt<16, 2, 8>(a, epi, p.grid, stream);
    case SK_K64_W10: return launch_mt<16, 2, 10>(a, epi, p.grid, stream);
    case SK_K64_W16: return launch_epi<16, 2, 16, true>(a, epi, p.grid, stream);  // E4M3 weights only (48 KiB stages)
    default: break;
  }
  if (a.wscale) return HWOCR_EINVAL;  // the 32-wide-K kernel has no E4M3 form (never planned for one)
  switch (epi) {
    case EPI_LINEAR: launch256<EPI_LINEAR>(a, p.grid, stream); break;
    case EPI_SWIGLU: launch256<EPI_SWIGLU>(a, p.grid, stream); break;
    case EPI_GEGLU: launch256<EPI_GEGLU>(a, p.grid, stream); break;
    case EPI_PARTIAL: launch256<EPI_PARTIAL>(a, p.grid, stream); break;
    default: return HWOCR_EINVAL;
  }
  return hwocr_launch_status();
}
