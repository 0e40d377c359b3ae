// Shared by the GEMM kernels: epilogue codes, activation helpers with the reference's bf16 rounding points, the
// argument block of the wide kernels and the per-tile epilogue store.
#pragma once
#include "common.h"
#include "hwocr.h"

namespace gemm {

enum : int {
  EPI_LINEAR = HWOCR_EPI_LINEAR,        // bf16(acc + bias)
  EPI_RESIDUAL = HWOCR_EPI_RESIDUAL,    // bf16(bf16(acc + bias) + res)
  EPI_QUICKGELU = HWOCR_EPI_QUICKGELU,  // x*sigmoid(1.702x), each step rounded like the bf16 module chain
  EPI_GELU = HWOCR_EPI_GELU,            // exact erf GELU of bf16(acc + bias)
  EPI_SWIGLU = HWOCR_EPI_SWIGLU,        // rows interleaved [16 gate][16 up]: bf16(bf16(silu(g)) * u)
  EPI_PARTIAL = HWOCR_EPI_PARTIAL,      // fp32 split-K slab (skinny only)
  EPI_GELU_TANH = HWOCR_EPI_GELU_TANH,  // tanh-approximated GELU of bf16(acc + bias)
  EPI_GEGLU = HWOCR_EPI_GEGLU,          // as EPI_SWIGLU with the tanh GELU on the gate
  EPI_VIT_QKV = 8                       // 256x256 kernel only (hwocr_gemm_vit_qkv): bias, vision rotary, head split, V transposed
};
template <int EPI> inline constexpr bool is_glu = EPI == EPI_SWIGLU || EPI == EPI_GEGLU;

// The activations run once per output element in the GEMM epilogues (128 elements per lane and tile), so their instruction
// count is tile time: an IEEE fp32 division is ~10 VALU instructions and tanhf ~40; here a sigmoid is exp + rcp (v_exp_f32,
// v_rcp_f32: 1 ulp each) and tanh is written through it.  Every result is rounded to bf16 (8 significant bits) right after.
__device__ __forceinline__ float fast_sigmoid(float t) { return __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }
__device__ __forceinline__ float act_quick_gelu(float v) {
  const float t = rbf(1.702f * v);
  const float s = rbf(fast_sigmoid(t));
  return v * s;
}
__device__ __forceinline__ float act_gelu_erf(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_silu(float v) { return v * fast_sigmoid(v); }
// torch's tanh GELU: 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3); 1 + tanh(u) = 2 sigmoid(2u)
__device__ __forceinline__ float act_gelu_tanh(float v) {
  const float u = 0.79788456080286535588f * (v + 0.044715f * (v * v * v));
  return v * fast_sigmoid(2.0f * u);
}
template <int EPI> __device__ __forceinline__ float glu_gate(float g) {
  if constexpr (EPI == EPI_GEGLU) return act_gelu_tanh(g);
  else return act_silu(g);
}

// ---- the same activations on PAIRS of outputs: v_pk_mul_f32 / v_pk_add_f32 do two fp32 operations per issued instruction (each lane
// half rounded on its own: the results are those of the scalar helpers above bit for bit; -ffp-contract=off keeps the products and sums
// apart).  An epilogue is 128-256 outputs per lane of vector work with no MFMA beside it: instructions issued are its time.
__device__ __forceinline__ f32x2 rbf2(f32x2 v) {
  const bf16x2 b = __builtin_convertvector(v, bf16x2);  // ONE v_cvt_pk_bf16_f32 (two scalar casts compile to two)
  const unsigned u = __builtin_bit_cast(unsigned, b);
  return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ f32x2 fast_sigmoid2(f32x2 t) {
  const f32x2 m = t * -1.44269504f;  // __expf(-t) = exp2(t * -log2(e)), as the scalar form compiles
  const f32x2 e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
  const f32x2 d = e + 1.0f;
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 act_quick_gelu2(f32x2 v) {
  const f32x2 t = rbf2(v * 1.702f);
  const f32x2 s = rbf2(fast_sigmoid2(t));
  return v * s;
}
__device__ __forceinline__ f32x2 act_silu2(f32x2 v) { return v * fast_sigmoid2(v); }
__device__ __forceinline__ f32x2 act_gelu_tanh2(f32x2 v) {
  const f32x2 u = (v + (v * v * v) * 0.044715f) * 0.79788456080286535588f;
  return v * fast_sigmoid2(u * 2.0f);
}
// bf16(act(bf16(acc + bias))) of a lane's four outputs (the staged epilogues of the 256 x 256 kernels)
template <int EPI>
__device__ __forceinline__ bf16x4 epi_act4(const f32x4& acc, const bf16x4& bias) {
  bf16x4 o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x2 v = f32x2{acc[2 * h], acc[2 * h + 1]} + f32x2{bf2f(bias[2 * h]), bf2f(bias[2 * h + 1])};
    if constexpr (EPI == EPI_QUICKGELU) v = act_quick_gelu2(rbf2(v));
    else if constexpr (EPI == EPI_GELU_TANH) v = act_gelu_tanh2(rbf2(v));
    else if constexpr (EPI == EPI_GELU) v = f32x2{act_gelu_erf(rbf(v[0])), act_gelu_erf(rbf(v[1]))};
    const bf16x2 ob = __builtin_convertvector(v, bf16x2);
    o[2 * h] = ob[0];
    o[2 * h + 1] = ob[1];
  }
  return o;
}

struct WideArgs {
  const bf16* X; const bf16* W; const bf16* bias; const bf16* res; bf16* out;
  int M, N, K, ldx, ldw, ldo, ldres, tilesM, tilesN;
  const float* xscale = nullptr;  // fp8 operands only (gemm256.hip): one scale per activation row / weight row
  const float* wscale = nullptr;
  hwocr_vit_split vs = {};        // EPI_VIT_QKV only: where the rotated / split / transposed result goes
};

// One 16x16 MFMA tile whose A operand was the weight tile: the lane holds out[m][n .. n+3] (4 consecutive features).
template <int EPI>
__device__ __forceinline__ void store_tile(const WideArgs& a, const f32x4& acc, int m, int n) {
  static_assert(!is_glu<EPI>, "gated epilogues consume a (gate, up) tile pair: store_glu");
  if (m >= a.M || n >= a.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r];
  if (a.bias) {
    const bf16x4 b = *(const bf16x4*)(a.bias + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bf2f(b[r]);
  }
  bf16x4 o;
  if constexpr (EPI == EPI_RESIDUAL) {
    const bf16x4 rs = *(const bf16x4*)(a.res + (size_t)m * a.ldres + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(v[r]) + bf2f(rs[r]));
  } else if constexpr (EPI == EPI_QUICKGELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(act_quick_gelu(rbf(v[r])));
  } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(act_gelu_erf(rbf(v[r])));
  } else if constexpr (EPI == EPI_GELU_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(act_gelu_tanh(rbf(v[r])));
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
  }
  *(bf16x4*)(a.out + (size_t)m * a.ldo + n) = o;
}

// gate tile (weight rows n_gate .. +15) and the up tile that follows it -> out[m][n_gate/2 + 4q .. +3]
template <int EPI>
__device__ __forceinline__ void store_glu(const WideArgs& a, const f32x4& g, const f32x4& u, int m, int n_gate, int q) {
  if (m >= a.M || n_gate >= a.N) return;
  float gb[4] = {0.f, 0.f, 0.f, 0.f}, ub[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {  // bias rows follow the interleaved weight rows (Qwen2.5-VL vision MLP)
    const bf16x4 bg = *(const bf16x4*)(a.bias + n_gate + 4 * q), bu = *(const bf16x4*)(a.bias + n_gate + 16 + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) { gb[r] = bf2f(bg[r]); ub[r] = bf2f(bu[r]); }
  }
  bf16x4 o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2 gg = rbf2(f32x2{g[2 * h], g[2 * h + 1]} + f32x2{gb[2 * h], gb[2 * h + 1]});
    const f32x2 uu = rbf2(f32x2{u[2 * h], u[2 * h + 1]} + f32x2{ub[2 * h], ub[2 * h + 1]});
    const f32x2 act = EPI == EPI_GEGLU ? act_gelu_tanh2(gg) : act_silu2(gg);
    const f32x2 y = rbf2(act) * uu;
    const bf16x2 ob = __builtin_convertvector(y, bf16x2);
    o[2 * h] = ob[0];
    o[2 * h + 1] = ob[1];
  }
  *(bf16x4*)(a.out + (size_t)m * a.ldo + (n_gate >> 1) + 4 * q) = o;
}

// tile of a linear id (ids with equal id % 8 share an XCD when consecutive workgroups are dealt round-robin): XCD-contiguous
// chunks, then GROUP row panels swept column-major (L2 reuse of both panels)
__device__ __forceinline__ void tile_of_id(int id, int tilesM, int tilesN, int group, int& tm, int& tn) {
  const int nwg = tilesM * tilesN;
  const int wg = xcd_remap(id, nwg);
  const int per_group = group * tilesN;
  const int g = wg / per_group, rem = wg - g * per_group;
  const int gm = min(group, tilesM - g * group);
  tm = g * group + rem % gm;
  tn = rem / gm;
}
__device__ __forceinline__ void tile_of_block(int tilesM, int tilesN, int group, int& tm, int& tn) {
  tile_of_id(blockIdx.x, tilesM, tilesN, group, tm, tn);
}

}  // namespace gemm

// decode-step GEMM over fragment-tiled weights (gemm_stream.hip); out is bf16 [Bsz][ldo] or, for EPI_PARTIAL, fp32
// slabs [splitk][Bsz][ldo]
struct StreamArgs {
  const bf16* X; const bf16* W; const bf16* bias; void* out;
  int Bsz, N, K, ldx, ldo, ktiles_per_slice;
  const float* wscale = nullptr;  // non-NULL: W holds E4M3 codes in the byte-tiled layout (hwocr_tile_weights_fp8), one scale per feature
};
int hwocr_gemm_stream(StreamArgs a, int epi, int splitk, hipStream_t stream);
// the kernel instance hwocr_gemm_stream would run for this shape (static string), without launching anything
int hwocr_gemm_stream_variant(int Bsz, int N, int K, int epi, int splitk, bool w8, const char** name);

// the fused vision QKV form is taken when the attention width is whole 256-wide tiles (no tile mixes q, k and v), the rotary
// pairs of a head fall into 16-byte pieces and there are enough rows for the 256x256 kernel
inline bool vit_qkv_fusable(int M, int heads, int hd) {
  return M >= 1024 && (heads * hd) % 256 == 0 && hd % 16 == 0 && hd <= 128;
}
// launcher of the 256x256 kernel (gemm256.hip); returns HWOCR_EINVAL when the shape does not qualify
int hwocr_gemm_wide256(const gemm::WideArgs& a, int epi, hipStream_t stream);
int hwocr_gemm_wide256_fp8(const gemm::WideArgs& a, int epi, hipStream_t stream);
int hwocr_gemm_wide256_vit_qkv(const gemm::WideArgs& a, bool fp8, hipStream_t stream);
// the four-wave form (gemm256w4.hip); false: shape / epilogue not covered, nothing launched
bool hwocr_gemm_wide256_w4(const gemm::WideArgs& b, int epi, bool forced, hipStream_t stream);

