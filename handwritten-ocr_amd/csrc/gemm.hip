// bf16 GEMM kernels for the page-read path (gfx950 / MI355X).
//
//   out[M][N] = epilogue( X[M][K] . W[N][K]^T )        X, W, out bf16; fp32 accumulation on MFMA
//
// W keeps the [out_features][in_features] layout of the checkpoints the reference loads
// (ocr_agent/tools.py:705-709 -> HF nn.Linear weights), so both operands are K-contiguous and every
// MFMA fragment is one 16-byte access.  The weight tile is the MFMA A operand and the activation tile
// the B operand: each lane then owns 4 consecutive output features of one row -> 8-byte stores.
//
// Two kernels:
//   gemm_wide   : 128x128x64 tiles, LDS-DMA staging (global_load_lds_dwordx4), XOR-swizzled LDS image,
//                 double buffered.  ViT blocks, merger and decoder prefill (MFMA-bound).
//   gemm_skinny : M <= 128 rows (the reads in flight during decode).  Weights stream HBM -> VGPR once,
//                 the activation slice is shared through LDS, optional split-K with fp32 partial slabs
//                 (HBM-bound).
//
// Epilogues reproduce the points where the reference's bf16 modules materialise a tensor
// (HF modeling_qwen2_vl.py:293-301 VisionMlp, :453-466 Qwen2MLP, :442-448 residual adds).
#include "gemm_common.h"
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace gemm;

namespace {

// ------------------------------------------------------------------------------------------------
// gemm_wide
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int WIDE_LDS = 4 * TILE_BYTES; // 2 buffers x (X tile + W tile)
constexpr int GROUP_M = 8;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;

  int tm, tn;
  tile_of_block(a.tilesM, a.tilesN, GROUP_M, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA staging plan: wave w fills rows 32w..32w+31 of both tiles, 8 rows (1 KiB) per instruction.
  // LDS image: row r at r*128 B, 16-byte chunk p holds logical chunk p ^ ((r>>1)&7) (conflict-free ds_read_b128);
  // the DMA destination is lane-linear, so the permutation is applied to the per-lane SOURCE address.
  const bf16* xsrc[4];
  const bf16* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 32 * w + 8 * i + (lane >> 3);
    const int lc = (lane & 7) ^ ((r >> 1) & 7);
    xsrc[i] = a.X + (size_t)min(m0 + r, a.M - 1) * a.ldx + lc * 8;
    wsrc[i] = a.W + (size_t)min(n0 + r, a.N - 1) * a.ldw + lc * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* xb = smem + buf * 2 * TILE_BYTES + (32 * w) * 128;
    char* wb = xb + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const void*)(xsrc[i] + kt * BK), LDS_PTR(xb + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(wsrc[i] + kt * BK), LDS_PTR(wb + i * 1024), 16, 0, 0);
    }
  };

  // wave grid 2(M) x 2(N): each wave owns a 64(m) x 64(n) block = 4x4 MFMA 16x16 tiles
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int sw = (c >> 1) & 7;  // rows of one fragment are base(16-aligned) + c
  const int nk = a.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* xb = smem + cur * 2 * TILE_BYTES;
    const char* wb = xb + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[4], xf[4];
      const int choff = ((kk * 4 + q) ^ sw) << 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf[t] = *(const bf16x8*)(wb + (wn * 64 + t * 16 + c) * 128 + choff);
        xf[t] = *(const bf16x8*)(xb + (wm * 64 + t * 16 + c) * 128 + choff);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane (c,q) of tile (nt,mt) holds out[m = ..+c][n = ..+4q .. 4q+3]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + c;
    if constexpr (is_glu<EPI>) {
#pragma unroll
      for (int nt = 0; nt < 4; nt += 2) store_glu<EPI>(a, acc[nt][mt], acc[nt + 1][mt], m, n0 + wn * 64 + nt * 16, q);
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) store_tile<EPI>(a, acc[nt][mt], m, n0 + wn * 64 + nt * 16 + 4 * q);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_skinny
// ------------------------------------------------------------------------------------------------
// Work split: grid.x = tiles of (16*NT*WAVES) weight rows, grid.y = K slices.  Each wave streams its 16*NT weight rows
// over the K slice in chunks of KC elements, one barrier per chunk.
struct SkinnyArgs {
  const bf16* X; const bf16* W; const bf16* bias; void* out;
  int Bsz, N, K, ldx, ldw, ldo, kslice;
};

// TILED: W is stored in MFMA-fragment order [N/16][K/32][lane = 16*(k/8 % 4) + n % 16][8] (hwocr_tile_weights), so every
// fragment load of a wave is one contiguous KiB and a wave streams one contiguous region; row-major W costs 16 rows x
// 64 B per instruction, which the memory system serves at about half the rate (measured: LM head 141 -> 90 us).
//
// Pipeline: weights HBM -> VGPR (non-temporal 16-byte loads), activations L2 -> LDS by LDS-DMA, two stages: the loads
// of chunk ci+1 are issued right after the barrier that publishes chunk ci and fly while chunk ci is multiplied.
// (Tried and dropped, numbers in DESIGN.md: a third stage with asm-issued VGPR loads - the register allocator moved
// the destinations before the data landed; everything through LDS-DMA - correct, but one workgroup per CU and the
// DMA issue rate made it 1.5x slower.)
template <int NB, int EPI, int NT, int WAVES, int KC, bool TILED>
__global__ __launch_bounds__(64 * WAVES) void gemm_skinny_kernel(SkinnyArgs a) {
  constexpr int KS = KC / 32;            // MFMA k-steps per chunk
  constexpr int CPR = KC / 8;            // 16-byte chunks per activation row
  constexpr int RPI = 64 / CPR;          // activation rows filled by one LDS-DMA instruction
  constexpr int XBYTES = NB * 16 * KC * 2;
  constexpr int NINST = NB * 16 / RPI / WAVES;  // LDS-DMA instructions per wave per chunk
  static_assert((NB * 16 / RPI) % WAVES == 0, "activation staging must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x [NB*16][KC] bf16; chunk p of row r holds p ^ (r & 15)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT * WAVES) + 16 * NT * w;
  const int ks = blockIdx.y;
  const int kbeg = ks * a.kslice, kend = min(a.K, kbeg + a.kslice);
  const int nchunk = (kend - kbeg + KC - 1) / KC;

  const bf16* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (TILED)
      wrow[t] = a.W + ((size_t)min((n0 >> 4) + t, (a.N >> 4) - 1) * (a.K >> 5) * 64 + lane) * 8;
    else
      wrow[t] = a.W + (size_t)min(n0 + 16 * t + c, a.N - 1) * a.ldw + 8 * q;
  }
  const bf16* xsrc[NINST];
  int xlc[NINST];
#pragma unroll
  for (int i = 0; i < NINST; ++i) {
    const int r = RPI * (w * NINST + i) + lane / CPR;
    xlc[i] = ((lane % CPR) ^ (r & 15)) * 8;
    xsrc[i] = a.X + (size_t)min(r, a.Bsz - 1) * a.ldx;
  }

  f32x4 acc[NT][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 wA[NT][KS], wB[NT][KS];
  auto issue = [&](bf16x8 (&wr)[NT][KS], int ci) {
    const int kc = kbeg + ci * KC;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kk = min(kc + 32 * s, a.K - 32);
        wr[t][s] = __builtin_nontemporal_load((const bf16x8*)(wrow[t] + (TILED ? (size_t)(kk >> 5) * 512 : (size_t)kk)));
      }
    char* xb = smem + (ci & 1) * XBYTES + (w * NINST) * 1024;
#pragma unroll
    for (int i = 0; i < NINST; ++i)
      __builtin_amdgcn_global_load_lds((const void*)(xsrc[i] + min(kc + xlc[i], a.K - 8)), LDS_PTR(xb + i * 1024), 16, 0, 0);
  };
  auto body = [&](int ci, bf16x8 (&cur)[NT][KS], bf16x8 (&nxt)[NT][KS]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk ci (weights + activations), issued one iteration ago
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(cur[t][s]));  // plain register data from here on
    __syncthreads();  // all waves: chunk ci visible, chunk ci-1 consumed -> the other buffer is free
    if (ci + 1 < nchunk) issue(nxt, ci + 1);
    const char* xb = smem + (ci & 1) * XBYTES;
    const int nks = min(KS, (kend - (kbeg + ci * KC)) >> 5);
    // activation fragments of k-step s+1 are fetched from LDS while k-step s is multiplied
    bf16x8 xf[2][NB];
    auto read_x = [&](bf16x8 (&dst)[NB], int s) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int r = 16 * b + c;
        dst[b] = *(const bf16x8*)(xb + r * (KC * 2) + (((4 * s + q) ^ (r & 15)) << 4));
      }
    };
    read_x(xf[0], 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < nks) {
        if (s + 1 < KS && s + 1 < nks) read_x(xf[(s + 1) & 1], s + 1);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[t][s], xf[s & 1][b], acc[t][b], 0, 0, 0);
      }
    }
  };

  issue(wA, 0);
  for (int ci = 0; ci < nchunk; ci += 2) {
    body(ci, wA, wB);
    if (ci + 1 < nchunk) body(ci + 1, wB, wA);
  }

  // lane (c,q): acc[t][b][r] = out[row 16b + c][n0 + 16t + 4q + r]
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int m = 16 * b + c;
    if (m >= a.Bsz) continue;
    if constexpr (is_glu<EPI>) {
      static_assert(!is_glu<EPI> || NT == 2, "gate/up tiles come in pairs");
      if (n0 < a.N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gte = rbf(acc[0][b][r]);
          const float up = rbf(acc[NT - 1][b][r]);
          o[r] = f2bf(rbf(glu_gate<EPI>(gte)) * up);
        }
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + (n0 >> 1) + 4 * q) = o;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + 4 * q;
        if (n >= a.N) continue;
        if constexpr (EPI == EPI_PARTIAL) {
          float* dst = (float*)a.out + ((size_t)ks * a.Bsz + m) * a.ldo + n;
          *(f32x4*)dst = acc[t][b];
        } else {
          bf16x4 o;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[t][b][r];
          if (a.bias) {
            const bf16x4 bb = *(const bf16x4*)(a.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bf2f(bb[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
          *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) = o;
        }
      }
    }
  }
}

template <int NB, int EPI, int NT, int WAVES, int KC, bool TILED>
void launch_skinny_one(const SkinnyArgs& a, int splitk, hipStream_t st) {
  constexpr int LDS = 2 * NB * 16 * KC * 2;
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<NB, EPI, NT, WAVES, KC, TILED>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    return true;
  }();
  (void)done;
  const int rows = 16 * NT * WAVES;
  dim3 grid((a.N + rows - 1) / rows, splitk), block(64 * WAVES);
  hipLaunchKernelGGL((gemm_skinny_kernel<NB, EPI, NT, WAVES, KC, TILED>), grid, block, LDS, st, a);
}
template <int NB, int EPI, int NT, int WAVES, int KC>
void launch_skinny_cfg(const SkinnyArgs& a, int splitk, bool tiled, hipStream_t st) {
  if (tiled) launch_skinny_one<NB, EPI, NT, WAVES, KC, true>(a, splitk, st);
  else launch_skinny_one<NB, EPI, NT, WAVES, KC, false>(a, splitk, st);
}

// KC: 256-element chunks while two activation buffers stay <= 64 KiB, else 128
template <int NB>
int launch_skinny(const SkinnyArgs& a, int epi, int splitk, bool tiled, hipStream_t st) {
  constexpr int KC = NB <= 4 ? 256 : 128;
  // few weight rows -> smaller workgroups so that the grid still covers the chip
  const long wg4 = (long)((a.N + 127) / 128) * splitk;
  switch (epi) {
    case EPI_LINEAR:
      if (wg4 >= 512) launch_skinny_cfg<NB, EPI_LINEAR, 2, 4, KC>(a, splitk, tiled, st);
      else launch_skinny_cfg<NB, EPI_LINEAR, 1, 2, KC>(a, splitk, tiled, st);
      break;
    case EPI_SWIGLU:
      if (wg4 >= 512) launch_skinny_cfg<NB, EPI_SWIGLU, 2, 4, KC>(a, splitk, tiled, st);
      else launch_skinny_cfg<NB, EPI_SWIGLU, 2, 2, KC>(a, splitk, tiled, st);
      break;
    case EPI_GEGLU:
      if (wg4 >= 512) launch_skinny_cfg<NB, EPI_GEGLU, 2, 4, KC>(a, splitk, tiled, st);
      else launch_skinny_cfg<NB, EPI_GEGLU, 2, 2, KC>(a, splitk, tiled, st);
      break;
    case EPI_PARTIAL:
      if (wg4 >= 512) launch_skinny_cfg<NB, EPI_PARTIAL, 2, 4, KC>(a, splitk, tiled, st);
      else launch_skinny_cfg<NB, EPI_PARTIAL, 1, 2, KC>(a, splitk, tiled, st);
      break;
    default: return HWOCR_EINVAL;
  }
  return hwocr_launch_status();
}

}  // namespace

// ---- optional in-stream timing of gemm_wide launches (bench.py roofline leg): HIP events recorded on the launch
// stream around every launch while enabled; a few microseconds of overhead per launch, no host sync.
namespace {
struct WideProfile {
  int on = 0;  // 1: time the bf16 wide launches, 2: the fp8 ones
  int n = 0;   // launches recorded since the last enable
  std::vector<hipEvent_t> ev;      // 2 per launch, created on demand and reused: no cap, no silently dropped launch
  std::vector<double> flops;
  std::mutex mu;                   // two host threads launch side by side when two batches are in flight (pipeline.LanePipeline)
  // reserves the event pair of the next recorded launch (created if this is the first time that many launches are recorded) and
  // books its FLOPs; false if HIP refuses
  bool reserve(hipEvent_t& a, hipEvent_t& b, double fl) {
    std::lock_guard<std::mutex> lock(mu);
    while ((int)ev.size() < 2 * (n + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return false;
      ev.push_back(e);
    }
    if ((int)flops.size() < n + 1) flops.resize(n + 1);
    a = ev[2 * n];
    b = ev[2 * n + 1];
    flops[n] = fl;
    ++n;
    return true;
  }
} g_prof;
}  // namespace

extern "C" int hwocr_profile_enable(int on) {
  g_prof.on = on;
  if (on) g_prof.n = 0;
  return HWOCR_OK;
}

// sums over the launches recorded since the last enable; the caller must have synchronised the stream
extern "C" int hwocr_profile_read(double* total_ms, double* total_flops, long* launches) {
  double ms = 0.0, fl = 0.0;
  for (int i = 0; i < g_prof.n; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return HWOCR_ELAUNCH;
    ms += t;
    fl += g_prof.flops[i];
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = g_prof.n;
  return HWOCR_OK;
}

extern "C" int hwocr_gemm_wide(const void* X, const void* W, const void* bias, const void* res, void* out,
                               int M, int N, int K, int ldx, int ldw, int ldo, int ldres, int epi,
                               hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (N % 8) != 0 || (ldx % 8) || (ldw % 8) || (ldo % 4))
    return HWOCR_EINVAL;
  if ((epi == EPI_SWIGLU || epi == EPI_GEGLU) && (N % 32) != 0) return HWOCR_EINVAL;
  if (epi == EPI_RESIDUAL && (!res || (ldres % 4))) return HWOCR_EINVAL;
  WideArgs a{(const bf16*)X, (const bf16*)W, (const bf16*)bias, (const bf16*)res, (bf16*)out,
             M, N, K, ldx, ldw, ldo, ldres, (M + BM - 1) / BM, (N + BN - 1) / BN};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool prof = g_prof.on == 1 && g_prof.reserve(ev0, ev1, 2.0 * M * (double)N * K);
  static const bool use256 = HWOCR_DIAG_ENV_INT("HWOCR_GEMM256", 3) != 0;
  if (use256 && M >= 1024 && N >= 256 && (ldo % 8) == 0 && (epi != EPI_RESIDUAL || (ldres % 8) == 0)) {
    if (prof) (void)hipEventRecord(ev0, stream);
    const int rc = hwocr_gemm_wide256(a, epi, stream);
    if (prof) {
      (void)hipEventRecord(ev1, stream);
    }
    return rc;
  }
  HWOCR_PLAN("gemm_wide_kernel<epi=%d> M=%d N=%d K=%d tiles=%d", epi, M, N, K, a.tilesM * a.tilesN);
  dim3 grid(a.tilesM * a.tilesN), block(256);
  static const bool attr_done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_QUICKGELU>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_GELU_TANH>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI_GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    return true;
  }();
  (void)attr_done;
  if (prof) (void)hipEventRecord(ev0, stream);
  switch (epi) {
    case EPI_LINEAR: hipLaunchKernelGGL(gemm_wide_kernel<EPI_LINEAR>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL(gemm_wide_kernel<EPI_RESIDUAL>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_QUICKGELU: hipLaunchKernelGGL(gemm_wide_kernel<EPI_QUICKGELU>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_GELU: hipLaunchKernelGGL(gemm_wide_kernel<EPI_GELU>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_SWIGLU: hipLaunchKernelGGL(gemm_wide_kernel<EPI_SWIGLU>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_GELU_TANH: hipLaunchKernelGGL(gemm_wide_kernel<EPI_GELU_TANH>, grid, block, WIDE_LDS, stream, a); break;
    case EPI_GEGLU: hipLaunchKernelGGL(gemm_wide_kernel<EPI_GEGLU>, grid, block, WIDE_LDS, stream, a); break;
    default: return HWOCR_EINVAL;
  }
  if (prof) {
    (void)hipEventRecord(ev1, stream);
  }
  return hwocr_launch_status();
}

extern "C" int hwocr_gemm_wide_fp8(const void* X8, const float* xscale, const void* W8, const float* wscale,
                                   const void* bias, const void* res, void* out, int M, int N, int K, int ldx, int ldw,
                                   int ldo, int ldres, int epi, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!X8 || !W8 || !xscale || !wscale || M <= 0 || N <= 0 || K <= 0 || (K % 128) || (N % 8) || (ldx % 16) || (ldw % 16) ||
      (ldo % 8))
    return HWOCR_EINVAL;
  if ((epi == EPI_SWIGLU || epi == EPI_GEGLU) && (N % 32) != 0) return HWOCR_EINVAL;
  if (epi == EPI_RESIDUAL && (!res || (ldres % 8))) return HWOCR_EINVAL;
  WideArgs a{(const bf16*)X8, (const bf16*)W8, (const bf16*)bias, (const bf16*)res, (bf16*)out,
             M, N, K, ldx, ldw, ldo, ldres, 0, 0, xscale, wscale};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool prof = g_prof.on == 2 && g_prof.reserve(ev0, ev1, 2.0 * M * (double)N * K);
  if (prof) (void)hipEventRecord(ev0, stream);
  const int rc = hwocr_gemm_wide256_fp8(a, epi, stream);
  if (prof) {
    (void)hipEventRecord(ev1, stream);
  }
  return rc;
}

// QKV projection of a vision block with rotary + head split + V transpose in the epilogue (hwocr.h); timed with the other wide
// GEMMs when the bench's profile is on
extern "C" int hwocr_gemm_vit_qkv(const void* X, const void* W, const void* bias, int M, int K, int ldx, int ldw,
                                  const float* xscale, const float* wscale, const hwocr_vit_split* sp, hipStream_t stream) {
  (void)hipGetLastError();
  if (!X || !W || !sp || !sp->Q || !sp->K || !sp->VT || !sp->pos_h || !sp->pos_w || !sp->cos_tab || !sp->sin_tab) return HWOCR_EINVAL;
  const bool fp8 = xscale != nullptr || wscale != nullptr;
  if (fp8 && (!xscale || !wscale)) return HWOCR_EINVAL;
  if (M <= 0 || K <= 0 || sp->heads <= 0 || !vit_qkv_fusable(M, sp->heads, sp->hd) || sp->tok_ld < M || (sp->tok_ld % 64)) return HWOCR_EINVAL;
  if (fp8 ? ((K % 128) || (ldx % 16) || (ldw % 16)) : ((K % BK) || (ldx % 8) || (ldw % 8))) return HWOCR_EINVAL;
  const int N = 3 * sp->heads * sp->hd;
  WideArgs a{(const bf16*)X, (const bf16*)W, (const bf16*)bias, nullptr, nullptr, M, N, K, ldx, ldw, 0, 0, 0, 0, xscale, wscale, *sp};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool prof = g_prof.on == (fp8 ? 2 : 1) && g_prof.reserve(ev0, ev1, 2.0 * M * (double)N * K);
  if (prof) (void)hipEventRecord(ev0, stream);
  const int rc = hwocr_gemm_wide256_vit_qkv(a, fp8, stream);
  if (prof) {
    (void)hipEventRecord(ev1, stream);
  }
  return rc;
}

// [N][K] row-major -> fragment-tiled copy for gemm_skinny (layout in the kernel comment).  16-byte granules.
__global__ __launch_bounds__(256) void tile_weights_kernel(const bf16* src, bf16* dst, int N, int K, int ldw) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte granule each
  const long total = (long)N * (K >> 3);
  if (gid >= total) return;
  const int lane = gid & 63;
  const long blk = gid >> 6;  // (tile, kstep)
  const int ksteps = K >> 5;
  const int tile = blk / ksteps, kstep = blk % ksteps;
  const int c = lane & 15, q = lane >> 4;
  *(bf16x8*)(dst + gid * 8) = *(const bf16x8*)(src + (size_t)(tile * 16 + c) * ldw + kstep * 32 + q * 8);
}

extern "C" int hwocr_tile_weights(const void* src, void* dst, int N, int K, int ldw, hipStream_t stream) {
  (void)hipGetLastError();
  if (N <= 0 || K <= 0 || (N % 16) || (K % 32) || (ldw % 8)) return HWOCR_EINVAL;
  const long total = (long)N * (K >> 3);
  hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     (const bf16*)src, (bf16*)dst, N, K, ldw);
  return hwocr_launch_status();
}

// Which path a skinny call takes: the streaming kernel (fragment-tiled weights) or gemm_skinny_kernel.  Shared by the
// launcher and by hwocr_gemm_skinny_variant so that the parity tests can ask what a given call would run.
namespace {
bool skinny_args_ok(int Bsz, int N, int K, int ldx, int ldw, int ldo, int epi, int splitk, bool has_bias) {
  if (Bsz <= 0 || Bsz > 256 || N <= 0 || K <= 0 || (K % 32) || (N % 16) || ((epi == EPI_SWIGLU || epi == EPI_GEGLU) && (N % 32)) ||
      (ldx % 8) || (ldw % 8) || (ldo % 4) || splitk < 1)
    return false;
  if (epi != EPI_PARTIAL && splitk != 1) return false;
  if ((epi == EPI_SWIGLU || epi == EPI_GEGLU) && has_bias) return false;
  return true;
}
bool skinny_takes_stream(int Bsz, int N, int K, int epi, int splitk, int w_tiled) {
  // fragment-tiled weights: the LDS-DMA streaming kernel (gemm_stream.hip)
  static const bool use_stream = HWOCR_DIAG_ENV_INT("HWOCR_GEMM_STREAM", 1) != 0;
  if (!(use_stream && w_tiled && Bsz <= 256 && (K % 64) == 0)) return false;
  const int ktiles = K / 64, per = (ktiles + splitk - 1) / splitk;
  // a plain linear over more than one round of 16-tile groups (the LM head) re-stages x once per group: the older
  // kernel's 128-row groups do that cheaper (LM head 2B: 90 us against 104)
  const bool many_rounds = epi == EPI_LINEAR && N / 16 > 16 * 256 && Bsz <= 128;  // (at 129+ rows the older kernel is slower)
  return (splitk - 1) * per < ktiles && !many_rounds;
}
int skinny_nb(int Bsz) {  // batch tiles of the gemm_skinny_kernel instance that serves Bsz rows
  const int nb = (Bsz + 15) / 16;
  return nb <= 4 ? nb : nb <= 6 ? 6 : nb <= 8 ? 8 : nb <= 12 ? 12 : 16;
}
}  // namespace

extern "C" int hwocr_gemm_skinny_variant(int Bsz, int N, int K, int epi, int splitk, int w_tiled, char* name, int name_len) {
  if (!name || name_len < 8 || !skinny_args_ok(Bsz, N, K, K, K, N, epi, splitk, false)) return HWOCR_EINVAL;
  if (w_tiled == 2) {  // E4M3 weights (hwocr_gemm_skinny_w8): always the streaming kernel
    if (K % 64) return HWOCR_EINVAL;
    const char* v = nullptr;
    const int rc = hwocr_gemm_stream_variant(Bsz, N, K, epi, splitk, true, &v);
    if (rc != HWOCR_OK) return rc;
    snprintf(name, name_len, "%s e4m3 epi=%d", v, epi);
    return HWOCR_OK;
  }
  if (skinny_takes_stream(Bsz, N, K, epi, splitk, w_tiled)) {
    const char* v = nullptr;
    const int rc = hwocr_gemm_stream_variant(Bsz, N, K, epi, splitk, false, &v);
    if (rc != HWOCR_OK) return rc;
    snprintf(name, name_len, "%s epi=%d", v, epi);
    return HWOCR_OK;
  }
  const long wg4 = (long)((N + 127) / 128) * splitk;
  snprintf(name, name_len, "gemm_skinny_kernel<NB=%d,%s,%s> epi=%d", skinny_nb(Bsz), wg4 >= 512 ? "4 waves" : "2 waves",
           w_tiled ? "tiled" : "rowmajor", epi);
  return HWOCR_OK;
}

extern "C" int hwocr_gemm_skinny(const void* X, const void* W, const void* bias, void* out, int Bsz, int N,
                                 int K, int ldx, int ldw, int ldo, int epi, int splitk, int w_tiled,
                                 hipStream_t stream) {
  (void)hipGetLastError();  // drop stale status left by other HIP users of this thread (e.g. event queries)
  if (!skinny_args_ok(Bsz, N, K, ldx, ldw, ldo, epi, splitk, bias != nullptr)) return HWOCR_EINVAL;
  if (hwocr_plan_on()) {
    char name[128];
    const int rc = hwocr_gemm_skinny_variant(Bsz, N, K, epi, splitk, w_tiled, name, sizeof(name));
    if (rc == HWOCR_OK) hwocr_plan_note("%s rows=%d N=%d K=%d splitk=%d", name, Bsz, N, K, splitk);
    return rc;
  }
  if (skinny_takes_stream(Bsz, N, K, epi, splitk, w_tiled))
    return hwocr_gemm_stream(StreamArgs{(const bf16*)X, (const bf16*)W, (const bf16*)bias, out, Bsz, N, K, ldx, ldo, 0},
                             epi, splitk, stream);
  // K slice per split: whole 256-element chunks
  int chunks = (K + 255) / 256;
  int per = (chunks + splitk - 1) / splitk;
  SkinnyArgs a{(const bf16*)X, (const bf16*)W, (const bf16*)bias, out, Bsz, N, K, ldx, ldw, ldo, per * 256};
  if ((splitk - 1) * a.kslice >= K) return HWOCR_EINVAL;  // an empty slice would leave its slab unwritten
  switch (skinny_nb(Bsz)) {
    case 1: return launch_skinny<1>(a, epi, splitk, w_tiled != 0, stream);
    case 2: return launch_skinny<2>(a, epi, splitk, w_tiled != 0, stream);
    case 3: return launch_skinny<3>(a, epi, splitk, w_tiled != 0, stream);
    case 4: return launch_skinny<4>(a, epi, splitk, w_tiled != 0, stream);
    case 6: return launch_skinny<6>(a, epi, splitk, w_tiled != 0, stream);
    case 8: return launch_skinny<8>(a, epi, splitk, w_tiled != 0, stream);
    case 12: return launch_skinny<12>(a, epi, splitk, w_tiled != 0, stream);
    default: return launch_skinny<16>(a, epi, splitk, w_tiled != 0, stream);
  }
}

// ---- decode GEMMs on E4M3 weights (BASELINE config 4): the codes of hwocr_quant_rows_fp8 re-laid in the order the streaming
// kernel's DMAs and fragment reads want them.  [N][K] bytes -> [N/16][K/64][64 lanes][16 B]: lane (c = lane & 15, q = lane >> 4)
// holds row 16 tile + c, k = 64 kt + 8 q + {0..7} and then k = 64 kt + 32 + 8 q + {0..7}.
__global__ __launch_bounds__(256) void tile_weights_fp8_kernel(const unsigned char* src, unsigned char* dst, int N, int K, int ldw) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte granule each
  const long total = (long)N * (K >> 4);
  if (gid >= total) return;
  const int lane = gid & 63;
  const long blk = gid >> 6;  // (tile, k tile)
  const int ktiles = K >> 6;
  const int tile = blk / ktiles, kt = blk % ktiles;
  const int c = lane & 15, q = lane >> 4;
  const unsigned char* row = src + (size_t)(tile * 16 + c) * ldw + kt * 64 + q * 8;
  uint2 lo = *(const uint2*)row, hi = *(const uint2*)(row + 32);
  *(uint4*)(dst + gid * 16) = uint4{lo.x, lo.y, hi.x, hi.y};
}

extern "C" int hwocr_tile_weights_fp8(const void* src, void* dst, int N, int K, int ldw, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || N <= 0 || K <= 0 || (N % 16) || (K % 64) || (ldw % 8)) return HWOCR_EINVAL;
  const long total = (long)N * (K >> 4);
  hipLaunchKernelGGL(tile_weights_fp8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     (const unsigned char*)src, (unsigned char*)dst, N, K, ldw);
  return hwocr_launch_status();
}

extern "C" int hwocr_gemm_skinny_w8(const void* X, const void* W8t, const float* wscale, const void* bias, void* out, int Bsz,
                                    int N, int K, int ldx, int ldo, int epi, int splitk, hipStream_t stream) {
  (void)hipGetLastError();
  if (!X || !W8t || !wscale || !out || (K % 64) || !skinny_args_ok(Bsz, N, K, ldx, K, ldo, epi, splitk, bias != nullptr)) return HWOCR_EINVAL;
  if (hwocr_plan_on()) {
    char name[128];
    const int rc = hwocr_gemm_skinny_variant(Bsz, N, K, epi, splitk, 2, name, sizeof(name));
    if (rc == HWOCR_OK) hwocr_plan_note("%s rows=%d N=%d K=%d splitk=%d", name, Bsz, N, K, splitk);
    return rc;
  }
  StreamArgs a{(const bf16*)X, (const bf16*)W8t, (const bf16*)bias, out, Bsz, N, K, ldx, ldo, 0};
  a.wscale = wscale;
  return hwocr_gemm_stream(a, epi, splitk, stream);
}
