// Argument block and LDS tile geometry shared by the attention translation units (attention.hip, attention_vit80x.hip).
#pragma once
#include "common.h"

namespace hwocr_attn {

constexpr float NEG_BIG = -1.0e30f;

struct PrefillArgs {
  const bf16* Q; const bf16* K; const bf16* VT; bf16* O; const int* lens;
  long q_seg, q_head, q_row;
  long k_seg, k_head, k_row;
  long v_seg, v_head, v_row;
  long o_seg, o_row;
  int group;          // query heads per kv head
  float scale_log2;   // softmax scale * log2(e)
  int kv_tiled;       // K / V^T in the fragment-tiled cache layout (common.h) instead of rows
  int heads, nseg, qblocks;  // 1-D grid decomposition (attn_vit80_kernel)
  const int* seg_off;        // packed ragged segments (hwocr_attn_varlen): first row of every segment, multiple of 4
  // Lazy running-max update of the two specialised kernels: the accumulators are rescaled only when some query's tile maximum
  // exceeds its running reference by more than `slack` (log2 units); until then the weights are exp2(s - m_ref) <= 2^slack
  // - still exact relative precision in bf16 / fp32 - and O / l is unchanged.  0 = rescale on every new maximum.
  float slack = 0.f;
  unsigned long long* stamps = nullptr;  // attention_vit80x.hip built with -DVIT80X_STAMPS (tools/bench_vit80x_stamps.py): cycle totals per wave
};

constexpr int V80_K0 = 64 * 128, V80_K1 = 64 * 32, V80_VT = 88 * 128;
constexpr int V80_STAGE = V80_K0 + V80_K1 + V80_VT;  // 21504 B


// attention_vit80x.hip (built with its own flags): the one-wave-per-SIMD form of the head_dim-80 tower attention for long segments
void launch_vit80x(const PrefillArgs& a, int grid, hipStream_t st);

}  // namespace hwocr_attn
