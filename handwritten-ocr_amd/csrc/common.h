// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the page-read path.
// Wave = 64 lanes everywhere; bf16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HWOCR_OK 0
#define HWOCR_EINVAL 1
#define HWOCR_ELAUNCH 2

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }   // RNE (v_cvt_pk_bf16_f32)
// value after a round trip through bf16: reproduces the places where the reference's
// bf16 modules materialise an intermediate tensor.
// The widening goes through integer bits on purpose: written as (float)(bf16)x, LLVM's contraction folds
// fpext(fptrunc(a*b)) + c into fma(a, b, c) and silently drops the rounding.
__device__ __forceinline__ float rbf(float x) {
  const bf16 b = (bf16)x;
  return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: hardware deals consecutive ids round-robin
// over the 8 XCDs; after the remap each XCD owns one contiguous chunk of logical ids, so tiles
// that share operand panels share an L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (orig >> 3);
}

// Fragment-tiled KV-cache layout (decode streams the cache as contiguous KiB blocks, exactly in MFMA operand order).
// Offsets are in elements inside one (read, kv head) region of ctx*128 elements; keys come in blocks of 32.
//   K : [block][tile t=0,1][d-step s=0..3][lane = 16*((d%32)/8) + c][8]   with tile row c <-> key 8(c>>2) + (c&3) + 4t
//   V^T: [block][d-tile 0..7][lane = 16*((key%32)/8) + d%16][8 keys]
__device__ __forceinline__ long kv_tiled_k(int key, int d) {
  const int kl = key & 31, t = (kl >> 2) & 1, c = ((kl >> 3) << 2) | (kl & 3);
  return (((((long)(key >> 5) * 2 + t) * 4 + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + c) << 3) + (d & 7);
}
__device__ __forceinline__ long kv_tiled_v(int d, int key) {
  const int kl = key & 31;
  return ((((long)(key >> 5) * 8 + (d >> 4)) * 64 + (kl >> 3) * 16 + (d & 15)) << 3) + (kl & 7);
}

// Switches of CONCLUDED A/B experiments (kernel-path choices whose outcome is recorded in DESIGN.md / docs/LAB_NOTEBOOK.md) exist only in
// the diagnostic build (-DHWOCR_DIAG, csrc/diag/, loaded by tools/ through build.use_diag_library()): there the variable is read
// once per process; in the product library the expression IS the default, no name is compiled in and no untested path can be
// selected from the environment.  The one switch the product library keeps is HWOCR_VIT80_KERNEL (attention.hip; read per call and
// walked by tests/test_ops_gpu.py in one process).
#ifdef HWOCR_DIAG
#include <stdlib.h>
#define HWOCR_DIAG_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#define HWOCR_DIAG_ENV_FLOAT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? (float)atof(e_) : (dflt); }())
#else
#define HWOCR_DIAG_ENV_INT(name, dflt) (dflt)
#define HWOCR_DIAG_ENV_FLOAT(name, dflt) (dflt)
#endif

// E4M3 KV cache of 256-wide heads (hwocr_kv.fp8): BYTE offsets inside one (read, kv head) region of ctx * 256 bytes, 32-key blocks of
// 8 KiB.  A lane's 16 bytes are its 8 codes of two consecutive k-steps (K) / d-tiles (V^T) of the decode attention's MFMA operands, so
// a block arrives as 4 + 4 (K: 2 tiles x 4 step pairs) + 8 (V^T: 8 tile pairs) 1-KiB loads per wave:
//   K  : [block][tile t = 0,1][step pair][lane = 16 qd + c][16 B]   tile row c <-> key 8 (c >> 2) + (c & 3) + 4 t, d = 64 pair + 32 half + 8 qd + e
//   V^T: [block][d-tile pair][lane = 16 qd + c][16 B]                row d = 32 pair + 16 half + c, keys 8 qd + e
__device__ __forceinline__ long kv8_k(int key, int d) {
  const int kl = key & 31, t = (kl >> 2) & 1, c = ((kl >> 3) << 2) | (kl & 3);
  const int s = d >> 5, qd = (d >> 3) & 3;
  return (((((long)(key >> 5) * 2 + t) * 4 + (s >> 1)) * 64 + qd * 16 + c) << 4) + (s & 1) * 8 + (d & 7);
}
__device__ __forceinline__ long kv8_v(int d, int key) {
  const int kl = key & 31;
  return ((((long)(key >> 5) * 8 + (d >> 5)) * 64 + (kl >> 3) * 16 + (d & 15)) << 4) + ((d >> 4) & 1) * 8 + (kl & 7);
}
// 8 E4M3 codes (two dwords) -> the bf16 fragment they stand for (every E4M3 value is exactly representable in bf16);
// v_cvt_scalef32_pk_bf16_fp8 (gfx950): two codes -> two packed bf16 per instruction, scale 1
__device__ __forceinline__ bf16x8 e4m3x8_bf16(int lo, int hi) {
  const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  return __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3), __builtin_shufflevector(c, d, 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
}

// last launch failure of this process (which launcher, which HIP error): read back through hwocr_last_error()
extern "C" void hwocr_record_error(const char* where, int hip_error, const char* text);
static inline int hwocr_launch_status_at(const char* where) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return HWOCR_OK;
  hwocr_record_error(where, (int)e, hipGetErrorString(e));
  return HWOCR_ELAUNCH;
}
#define hwocr_launch_status() hwocr_launch_status_at(__func__)

// CUs the calling thread's launches may count on (hwocr_set_cu_budget, runtime.hip); 0 = the whole device
int hwocr_cu_budget();

// Plan recording (hwocr_plan_begin / hwocr_plan_end, runtime.hip): while it is on for the calling thread every launcher
// validates its arguments as usual, notes the kernel instance and geometry it WOULD launch and returns HWOCR_OK without touching
// the device.  hwocr_vit_forward / hwocr_prefill run under it with placeholder pointers give the launch list of a configuration —
// what tests/test_wide_variants.py holds against the parity cases (no GPU needed).
bool hwocr_plan_on();
void hwocr_plan_note(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#define HWOCR_PLAN(...)            \
  do {                             \
    if (hwocr_plan_on()) {         \
      hwocr_plan_note(__VA_ARGS__); \
      return HWOCR_OK;             \
    }                              \
  } while (0)
