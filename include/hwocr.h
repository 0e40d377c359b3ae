/*
 * hwocr.h — C ABI of the MI355X-native page-read engine (libhwocr_hip.so) and of the native text
 * kernels (libhwocr_text.so).
 *
 * The reference (marwanbounassif/handwritten-ocr) has no native boundary: its hot path is the Python call
 *   ocr_agent/tools.py:728  run_ocr(image_path, params) -> str
 * which drives HF transformers (processor -> model.generate -> decode, tools.py:756-769), plus the pure-Python
 * string DP of compare_versions / merge_versions (tools.py:326-493).  This header is the seam a maintainer binds
 * instead (ctypes stub in INTEGRATION.md): plain device pointers, sizes and a HIP stream; no torch types;
 * every function returns HWOCR_OK (0) or an error code and never throws; no hidden allocation — the caller owns
 * every buffer, including workspaces.
 *
 * All device tensors are bf16 unless stated.  "rows" are tokens (vision patches or prompt positions) or,
 * during decode, the reads in flight (one row per read).
 */
#ifndef HWOCR_H
#define HWOCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hwocr_stream_t; /* == hipStream_t */

#define HWOCR_OK 0
#define HWOCR_EINVAL 1
#define HWOCR_ELAUNCH 2

/* GEMM epilogues (see csrc/gemm.hip) */
#define HWOCR_EPI_LINEAR 0
#define HWOCR_EPI_RESIDUAL 1
#define HWOCR_EPI_QUICKGELU 2
#define HWOCR_EPI_GELU 3
#define HWOCR_EPI_SWIGLU 4
#define HWOCR_EPI_PARTIAL 5
#define HWOCR_EPI_GELU_TANH 6 /* nn.GELU(approximate="tanh"): SigLIP MLP (HF siglip/modeling_siglip.py:310-322) */
#define HWOCR_EPI_GEGLU 7     /* SWIGLU's interleaved tile pairs with the tanh GELU as gate: Gemma MLP (HF gemma/modeling_gemma.py:84-97) */

#define HWOCR_ABI_VERSION 14 /* bumped whenever a signature or struct layout below changes */
int hwocr_abi_version(void);
/* text of the most recent launch failure in this process ("" if none): launcher name + HIP error */
const char* hwocr_last_error(void);

/* ---- single operators (each replaces the ATen op reached from the cited HF module) ------------------------- */

/* out[M][N] = epi(X[M][K] . W[N][K]^T); replaces nn.Linear / Conv3d-as-GEMM (HF modeling_qwen2_vl.py:266-274,
 * :293-301, :349-350, :453-466, :501-504).  K % 64 == 0, N % 8 == 0. */
int hwocr_gemm_wide(const void* X, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                    int ldx, int ldw, int ldo, int ldres, int epi, hwocr_stream_t stream);

/* E4M3 path of the wide GEMM (BASELINE config 4: "fp8 MFMA on CDNA4"; no reference counterpart — HF runs bf16).
 * hwocr_quant_rows_fp8: per row, scale = max|x| / 448 (1 for an all-zero row), Q = e4m3(x * (448 / max|x|)) with
 * round-to-nearest-even; used for activations (per token) and, once at load time, for weights (per output feature).
 * hwocr_gemm_wide_fp8: out = epi(xscale[m] * wscale[n] * sum_k X8[m][k] W8[n][k]) on v_mfma_f32_16x16x128_f8f6f4 with fp32
 * accumulation; bias / res / out stay bf16 and the epilogues are those of hwocr_gemm_wide.  K % 128 == 0, N % 8 == 0. */
int hwocr_quant_rows_fp8(const void* X, void* Q, float* scale, int rows, int K, int ldx, int ldq, hwocr_stream_t stream);
/* hwocr_layernorm / hwocr_add_rmsnorm (no slabs, no gather) that hand their bf16 result straight to hwocr_quant_rows_fp8
 * inside the kernel: same codes and scales as the two calls, the bf16 row never goes to HBM */
int hwocr_layernorm_fp8(const void* x, const void* w, const void* b, void* q8, float* q8s, int rows, int D, int ldx, int ldq,
                        float eps, hwocr_stream_t stream);
int hwocr_rmsnorm_fp8(const void* h, int ldh, const void* w, void* q8, float* q8s, int ldq, int rows, int D, float eps,
                      int gemma, hwocr_stream_t stream);
int hwocr_gemm_wide_fp8(const void* X8, const float* xscale, const void* W8, const float* wscale, const void* bias,
                        const void* res, void* out, int M, int N, int K, int ldx, int ldw, int ldo, int ldres, int epi,
                        hwocr_stream_t stream);

/* Same contraction for <= 256 rows (decode).  epi PARTIAL writes fp32 slabs out[splitk][Bsz][ldo].
 * w_tiled != 0: W is the fragment-tiled copy made by hwocr_tile_weights (ldw ignored). */
int hwocr_gemm_skinny(const void* X, const void* W, const void* bias, void* out, int Bsz, int N, int K, int ldx,
                      int ldw, int ldo, int epi, int splitk, int w_tiled, hwocr_stream_t stream);

/* The decode GEMMs on E4M3 weights (BASELINE config 4, "fp8 ... on CDNA4"; no reference counterpart — HF runs bf16).  A decode
 * step at <= 256 rows is bound by the bytes entering the compute units, not by the matrix pipes, so the weights are STORED as E4M3
 * codes + one fp32 scale per output feature (hwocr_quant_rows_fp8) and multiplied as the bf16 values they stand for (every E4M3
 * value is exact in bf16) against the bf16 activations: out = epi(wscale[n] * sum_k X[m][k] * e4m3(W8[n][k])), fp32 accumulation —
 * half the weight bytes of hwocr_gemm_skinny, no activation quantisation.  hwocr_tile_weights_fp8: [N][K] code bytes ->
 * [N/16][K/64][64 lanes][16 B], the order of the streaming kernel's DMAs (N % 16 == 0, K % 64 == 0).  epi / splitk as
 * hwocr_gemm_skinny (PARTIAL slabs carry the scale already). */
int hwocr_tile_weights_fp8(const void* src, void* dst, int N, int K, int ldw, hwocr_stream_t stream);
int hwocr_gemm_skinny_w8(const void* X, const void* W8t, const float* wscale, const void* bias, void* out, int Bsz, int N, int K,
                         int ldx, int ldo, int epi, int splitk, hwocr_stream_t stream);

/* The decode GEMMs with AT MOST 16 reads in flight (one page = 3 reads; the tail of a continuous batch): csrc/gemm_rows16.hip.
 * out[Bsz <= 16][N] = x[Bsz][K] . Wt^T over the fragment-tiled bf16 weights (hwocr_tile_weights), no K-tile ring: the waves of a
 * workgroup split a tile's K range, stream their slices HBM -> registers and sum them through LDS in ascending K order.
 *   norm == NULL: x = X (bf16 rows, ldx).   norm != NULL (K <= 4096): the RMSNorm in front of the projection is the kernel's
 *   prologue — x = RMSNorm(h'), h' = bf16(bf16(sum of norm->nslab <= 4 fp32 slabs) + h_in) (nslab == 0: h' = h_in), with the
 *   reference's rounding chain (hwocr_add_rmsnorm's arithmetic; gemma: (1 + w) in fp32); h' is stored once to h_out (may be NULL;
 *   must NOT alias h_in: every workgroup reads h_in).
 *   epi: HWOCR_EPI_PARTIAL (fp32 slabs [splitk][Bsz][ldo], the only one that takes splitk > 1), HWOCR_EPI_RESIDUAL (out is the
 *   bf16 residual stream, updated in place: out <- bf16(bf16(acc) + out)), HWOCR_EPI_SWIGLU / _GEGLU (interleaved gate / up tiles,
 *   out [Bsz][N/2]), HWOCR_EPI_LINEAR (bf16, no bias).  N % 16 == 0 (gated: % 32), K % 32 == 0. */
typedef struct hwocr_rows16_norm {
  const void* h_in; void* h_out; int ldh;
  const float* slabs; int nslab; long slab_stride; int ld_slab;
  const void* norm_w; float eps; int gemma;
} hwocr_rows16_norm;
int hwocr_gemm_rows16(const void* X, int ldx, const void* Wt, void* out, int ldo, int Bsz, int N, int K, int epi, int splitk,
                      const hwocr_rows16_norm* norm, hwocr_stream_t stream);

/* Name of the kernel instance the call hwocr_gemm_skinny(Bsz, N, K, epi, splitk, w_tiled) would launch (ldx = ldw = K,
 * ldo = N; w_tiled == 2: hwocr_gemm_skinny_w8), written NUL-terminated into name[name_len]; launches nothing and needs no device.  The decode GEMMs pick among
 * several instances by shape (csrc/gemm_stream.hip plan_stream): the parity tests use this to prove that every instance a
 * shipped configuration dispatches has an oracle case. */
int hwocr_gemm_skinny_variant(int Bsz, int N, int K, int epi, int splitk, int w_tiled, char* name, int name_len);

/* [N][K] row-major weights -> [N/16][K/32][64 lanes][8] (lane = 16*((k/8)%4) + n%16): the order in which one wave's
 * MFMA A fragments are consumed, so decode streams every weight byte as contiguous KiB blocks.  N%16 == 0, K%32 == 0. */
int hwocr_tile_weights(const void* src, void* dst, int N, int K, int ldw, hwocr_stream_t stream);

/* Whole-segment attention (HF modeling_qwen2_vl.py:375-418 vision, :553-569 decoder prefill).
 * Element strides; V is passed transposed (VT[d][key]); every segment's key range must be readable up to the
 * next multiple of 64. */
int hwocr_attn_prefill(const void* Q, const void* K, const void* VT, void* O, const int* lens, int nseg, int heads,
                       int group, int head_dim, int max_len, int causal, long q_seg, long q_head, long q_row,
                       long k_seg, long k_head, long k_row, long v_seg, long v_head, long v_row, long o_seg,
                       long o_row, float scale, int kv_tiled, hwocr_stream_t stream);

/* Packed ragged segments, non-causal, one kv head per query head: segment s owns rows seg_off[s] .. +lens[s] of the
 * shared [rows] axis (seg_off % 4 == 0).  The windowed layers of the Qwen2.5-VL vision tower
 * (HF modeling_qwen2_5_vl.py:262-285 with cu_window_seqlens).  VT must be readable 64 keys past the last segment start. */
int hwocr_attn_varlen(const void* Q, const void* K, const void* VT, void* O, const int* seg_off, const int* lens,
                      int nseg, int heads, int head_dim, int max_len, long q_head, long q_row, long k_head,
                      long k_row, long v_head, long v_row, long o_row, float scale, hwocr_stream_t stream);

/* One query token per read against its KV cache (HF modeling_qwen2_vl.py:553-569 with q_len == 1).  head_dim 128, or 256
 * (Gemma) with the row layout only.  kv_tiled (here and in the cache writers below): the cache is in the fragment-tiled
 * layout, strides k_row/v_row unused.  1 <= nsplit <= 16 (part_o / part_ml hold 16 splits).  nsplit > 1: the key range of a
 * (read, kv head) is split over nsplit workgroups whose partial results are merged — by the workgroup that arrives last, if
 * `arrive` gives it one int32 counter per (read, kv head) [nseq * Hkv] that is ZERO when the call is made (the kernel leaves it
 * zero), else (arrive == NULL) by a second launch.  Same arithmetic in the same order either way: identical bytes. */
int hwocr_attn_decode(const void* Q, const void* K, const void* VT, const int* lens, void* out, float* part_o,
                      float* part_ml, int* arrive, int nseq, int Hq, int Hkv, int nsplit, long k_seq, long k_head, long v_seq,
                      long v_head, long v_row, float scale, int head_dim, int kv_tiled, hwocr_stream_t stream);

/* hwocr_decode_qkv_finish (below) and hwocr_attn_decode in ONE launch: every (read, kv head) workgroup first sums the slabs of
 * its query heads, its key and its value head, rotates, appends k / v at cache slot lens - 1, then attends over lens keys.
 * Same arithmetic in the same order: outputs and cache bit-identical to the two calls.  The Q buffer is not used.  Reads outside
 * the invariants of hwocr_decode_qkv_finish are skipped as there (no output, HWOCR_STATUS_BAD_POSITION raised in *status). */
int hwocr_attn_decode_qkv(const float* slabs, int nslab, long slab_stride, const void* bias, void* K, void* VT,
                          const int* lens, const int* rope_delta, const void* cos_tab, const void* sin_tab, void* out,
                          float* part_o, float* part_ml, int* arrive, int nseq, int Hq, int Hkv, int nsplit, long k_seq, long k_head,
                          long v_seq, long v_head, long v_row, float scale, int head_dim, int kv_tiled, int ctx, int max_pos,
                          int* status, hwocr_stream_t stream);

/* hwocr_attn_decode_qkv over an E4M3 cache (hwocr_kv.fp8; head_dim 256, row-free operand order): the appended token is quantised
 * (its own scale), keys / values are converted to bf16 in registers - exact: E4M3 is a subset of bf16 - the key scale multiplies the
 * score, the value scale the softmax weight.  k_scale / v_scale: this layer's [seq][Hkv][ctx]. */
int hwocr_attn_decode_qkv_fp8kv(const float* slabs, int nslab, long slab_stride, const void* bias, void* K8, void* VT8,
                                float* k_scale, float* v_scale, const int* lens, const int* rope_delta, const void* cos_tab,
                                const void* sin_tab, void* out, float* part_o, float* part_ml, int* arrive, int nseq, int Hq, int Hkv,
                                int nsplit, float scale, int ctx, int max_pos, int* status, hwocr_stream_t stream);

/* bf16 K [nseq][Hkv] rows of 256 (strides k_seq / k_head, in elements) and V^T [nseq][Hkv][256][...] (v_seq / v_head / v_row) of
 * `keys` positions per read (a multiple of 32) -> the E4M3 cache regions K8 / VT8 [nseq][Hkv][ctx * 256 bytes] + scales
 * [nseq][Hkv][ctx] (hwocr_kv.fp8 layout).  The prefill's cache fill. */
int hwocr_kv_quant_fp8(const void* K, const void* VT, long k_seq, long k_head, long v_seq, long v_head, long v_row, void* K8, void* VT8,
                       float* k_scale, float* v_scale, int nseq, int Hkv, int keys, int ctx, hwocr_stream_t stream);

/* as hwocr_gemm_skinny_variant, for hwocr_attn_decode */
int hwocr_attn_decode_variant(int nsplit, int head_dim, int kv_tiled, char* name, int name_len);

/* uint8 HWC resized pages -> bf16 patch rows (HF image_processing_pil_qwen2_vl.py:152-187, :226-229).
 * row_src (optional, device int32 [gh*gw]): output row r of every image shows patch row_src[r] of the processor's
 * merge-block-major order — the window permutation of the Qwen2.5-VL tower (HF modeling_qwen2_5_vl.py:441-444). */
int hwocr_patchify(const void* img, const void* lut, void* out, int nimg, int H, int W, int patch, int merge,
                   int tps, int kpad, int rows_per_img_ld, const int* row_src, hwocr_stream_t stream);

int hwocr_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int D, int ldx, int ldo,
                    float eps, hwocr_stream_t stream);

/* h <- bf16(bf16(sum slabs + bias) + h) (if nslab > 0); out <- RMSNorm(h) * w (if out != NULL). */
int hwocr_add_rmsnorm(const float* slabs, int nslab, long slab_stride, int ld_slab, const void* bias, void* h,
                      int ldh, const void* w, void* out, int ldo, const int* row_index, int rows, int D, float eps,
                      int gemma, hwocr_stream_t stream);

/* Vision rotary + head split (HF modeling_qwen2_vl.py:239-248, :375-400): qkv [tokens][3][heads][hd] -> Q, K
 * [heads][tok_ld][hd] rotated (fp32 math on the bf16 values, one rounding) and V^T [heads][hd][tok_ld].  interleaved != 0: the
 * q / k features of a head arrive as rotary pairs side by side, [d0, d0 + hd/2, d1, d1 + hd/2, ...] — the row order
 * hwocr_gemm_vit_qkv wants of the QKV weight (hwocr_vit.qk_interleaved); the outputs are in the true feature order either way. */
int hwocr_vit_rope_split(const void* qkv, void* Q, void* K, void* VT, const int* pos_h, const int* pos_w,
                         const float* cos_tab, const float* sin_tab, int tokens, int tok_ld, int heads, int hd,
                         int interleaved, hwocr_stream_t stream);

/* The QKV projection of a vision block with hwocr_vit_rope_split folded into its epilogue: out = X . W^T + bias with W
 * [3 * heads * hd][K] whose q / k rows are pair-interleaved per head (above); every 256 x 256 output tile is rounded to bf16,
 * rotated / transposed in registers + LDS and stored straight into Q, K, V^T — the [tokens][3 * heads * hd] intermediate
 * never exists.  Bit-identical to hwocr_gemm_wide + hwocr_vit_rope_split(interleaved = 1).  xscale / wscale both non-NULL: X
 * and W hold E4M3 bytes with per-row scales (hwocr_gemm_wide_fp8).  Returns HWOCR_EINVAL unless M >= 1024,
 * heads * hd % 256 == 0 and hd % 16 == 0 (the caller then runs the two separate calls). */
typedef struct {
  void *Q, *K, *VT;
  const int *pos_h, *pos_w;        /* [tokens] */
  const float *cos_tab, *sin_tab;  /* fp32 [positions][hd / 4] */
  int heads, hd, tok_ld;           /* tok_ld: row pitch of Q / K (in tokens) and of V^T (in elements), >= M, % 64 == 0 */
} hwocr_vit_split;
int hwocr_gemm_vit_qkv(const void* X, const void* W, const void* bias, int M, int K, int ldx, int ldw, const float* xscale,
                       const float* wscale, const hwocr_vit_split* split, hwocr_stream_t stream);

/* rows are laid out [nseq][rows_per_seq]; row r belongs to read r / rows_per_seq (K, VT point at the first read's
 * cache), cache slot r % rows_per_seq.  head_dim 128 or 256; rope tables [maxpos][head_dim/2]; sec0 >= head_dim/2 turns
 * the three-axis M-RoPE into plain RoPE on pos[0] (Gemma) */
int hwocr_mrope_kv_prefill(const void* qkv, void* Q, void* K, void* VT, const int* pos, const void* cos_tab,
                           const void* sin_tab, int rows, int rows_per_seq, int Hq, int Hkv, int sec0, int sec1,
                           long k_seq, long k_head, long v_seq, long v_head, long v_row, int head_dim, int kv_tiled,
                           hwocr_stream_t stream);

/* Decode form: one row per read = the split-K slabs of the fused QKV projection (+ bias); rotary at position
 * lens - 1 + rope_delta, K / V^T appended at cache slot lens - 1.  The caller keeps 1 <= lens <= ctx and
 * 0 <= lens - 1 + rope_delta < max_pos (rows of the rope tables); a read that breaks this is skipped (nothing written,
 * nothing read outside the tables) and HWOCR_STATUS_BAD_POSITION is OR-ed into *status (device int, may be NULL). */
#define HWOCR_STATUS_BAD_POSITION 1
int hwocr_decode_qkv_finish(const float* slabs, int nslab, long slab_stride, const void* bias, void* Q, void* K,
                            void* VT, const int* lens, const int* rope_delta, const void* cos_tab,
                            const void* sin_tab, int nseq, int Hq, int Hkv, long k_seq, long k_head, long v_seq,
                            long v_head, long v_row, int head_dim, int kv_tiled, int ctx, int max_pos, int* status,
                            hwocr_stream_t stream);

int hwocr_embed_splice(const int* ids, const int* img_row, const void* table, const void* img, void* out, int rows,
                       int D, float scale, hwocr_stream_t stream);

/* seen (optional): bitmap [nseq][seen_ld 32-bit words] of the token ids already in a read's prompt + output; with
 * rep_penalty != 1 their fp32 scores are divided (positive) or multiplied (negative) by it before the argmax — HF
 * RepetitionPenaltyLogitsProcessor, which the Qwen2.5-VL / olmOCR generation configs switch on.  The caller fills in the
 * prompt's ids; every later call first adds the id it was fed (cur_ids on entry, when n_gen > 0).
 * split_ws (optional): device int32 [nseq][HWOCR_SELECT_WS_INTS] whose first int per read is ZERO when the call is made (the kernel
 * leaves it zero).  With it and nseq <= 16 a row is scanned by 16 workgroups instead of one and the last to arrive finishes the read;
 * the token picked is the same (maximum value, lowest index). */
#define HWOCR_SELECT_WS_INTS 40
int hwocr_argmax_advance(const void* logits, int ldl, int V, int nseq, int* cur_ids, int* lens, int* n_gen,
                         int* finished, int* out_tokens, int max_new, int min_new, const int* eos, int n_eos,
                         int pad_id, unsigned* seen, int seen_ld, float rep_penalty, int* split_ws, hwocr_stream_t stream);

/* generate(do_sample=True): temperature -> top-k -> top-p -> one multinomial draw per read, then the same bookkeeping as
 * hwocr_argmax_advance (replaces HF generation/logits_process.py Temperature / TopK / TopP warpers and the softmax +
 * torch.multinomial of generation/utils.py:_sample as reached from /root/reference/ocr_agent/tools.py:765 when the checkpoint's
 * generation_config.json says do_sample).  The procedure is exact integer arithmetic ("hwocr sampling v1", spelled out in
 * oracle/sampling.py and DESIGN.md): fixed-point weights, radix-selected thresholds (ties at a threshold are kept whole),
 * Philox4x32-10 keyed by `seed` with counter (read_ids[read] - or the row index when NULL -, step): a read's draws do not depend
 * on batch layout.
 * top_k <= 0 or >= V: off; top_p >= 1: off; temperature > 0.  debug (optional): [nseq][8] uint64 = max bits, top-k key, total mass,
 * nucleus target mass, nucleus key, kept mass, draw target, token. */
int hwocr_sample_advance(const void* logits, int ldl, int V, int nseq, int* cur_ids, int* lens, int* n_gen, int* finished,
                         int* out_tokens, int max_new, int min_new, const int* eos, int n_eos, int pad_id, unsigned* seen,
                         int seen_ld, float rep_penalty, float temperature, int top_k, float top_p, unsigned long long seed,
                         const int* read_ids, unsigned long long* debug, hwocr_stream_t stream);

/* ---- model-level entry points (what run_ocr's model.generate expands to) ----------------------------------- */

#define HWOCR_VIT_QWEN2 0   /* LayerNorm, fc1 -> QuickGELU -> fc2 (HF modeling_qwen2_vl.py:421-437) */
#define HWOCR_VIT_QWEN2_5 1 /* RMSNorm, biased gate/up/down SiLU MLP, windowed attention (HF modeling_qwen2_5_vl.py:293-322) */
#define HWOCR_VIT_SIGLIP 2  /* PaliGemma tower: biased patch conv + learned positions, LayerNorm, fc1 -> tanh GELU -> fc2, no rotary,
                             * post-LayerNorm, one linear projector (HF siglip/modeling_siglip.py:116-356, paligemma/modeling_paligemma.py:90-98) */

/* E4M3 copy of one weight matrix, [N][K] bytes + one fp32 scale per row (hwocr_quant_rows_fp8 of the bf16 matrix);
 * w == NULL: that GEMM stays in bf16 */
typedef struct { const void* w; const float* scale; } hwocr_w8;

typedef struct {
  /* QWEN2_5: ln*_b unused; fc1_w/fc1_b = gate_proj/up_proj rows interleaved in 16-row tiles [2*mlp_dim][dim] (+ bias
   * likewise), fc2 = down_proj [dim][mlp_dim]; mlp_dim is the intermediate size zero-padded to a multiple of 64 */
  const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  int windowed; /* QWEN2_5: attention inside windows (layer not in fullatt_block_indexes) */
  hwocr_w8 qkv8, proj8, fc18, fc28; /* optional E4M3 copies of qkv_w / proj_w / fc1_w / fc2_w (used when K % 128 == 0) */
} hwocr_vit_block;

typedef struct {
  int depth, dim, heads, mlp_dim, patch, merge, tps, kpad, out_dim, kind;
  int head_pad; /* 0, or the width heads are zero-padded to in Q/K/V^T and in the rows of qkv_w / columns of proj_w (SigLIP:
                 * head_dim 72 -> 80, scores and outputs unchanged; softmax scale stays (dim/heads)^-1/2) */
  float eps;
  const void* patch_w;            /* [dim][kpad], zero beyond 3*tps*patch^2 */
  const hwocr_vit_block* blocks;  /* host array[depth] of device pointers */
  const void *merger_ln_w, *merger_ln_b, *merger_fc1_w, *merger_fc1_b, *merger_fc2_w, *merger_fc2_b;
  const float *rope_cos, *rope_sin; /* fp32 [maxpos][head_dim/4] */
  const void* pixel_lut;            /* bf16 [3][256] */
  const void *patch_b, *pos_embed;  /* SIGLIP: conv bias [dim], learned positions [patches][dim]; else NULL */
  int qk_interleaved;               /* the q / k rows of every block's qkv_w / qkv_b (and qkv8) are pair-interleaved per head */
} hwocr_vit;

typedef struct { /* all device buffers, rows = nimg * rows_per_img_ld; vt holds 64 elements of slack past rows*dim */
  void *patches, *x, *xn, *qkv, *q, *k, *vt, *attn, *mlp, *merge_mid;
  void* q8;   /* E4M3 staging of one GEMM input, rows * max(dim, heads*head width, mlp_dim) bytes; NULL without fp8 weights */
  float* q8s; /* its row scales [rows] */
} hwocr_vit_ws;

typedef struct { /* device index tables of one batch of equally sized pages */
  const int *pos_h, *pos_w; /* [rows] patch coordinates in buffer-row order */
  const int *seg_lens;      /* [nimg] real patches per page */
  const int *row_src;       /* [gh*gw] window permutation handed to hwocr_patchify, or NULL */
  const int *win_off, *win_lens; /* [nwin] windows of the whole batch as row ranges (QWEN2_5), or NULL */
  int nwin, max_win;
} hwocr_vit_layout;

/* images: uint8 [nimg][H][W][3] already resized to multiples of patch*merge; out: [rows/merge^2][out_dim], in buffer-row
 * order (with row_src: window order — the caller's splice table undoes it, HF modeling_qwen2_5_vl.py:474-476) */
int hwocr_vit_forward(const hwocr_vit* m, const hwocr_vit_ws* ws, const void* images, int nimg, int H, int W,
                      int rows_per_img_ld, const hwocr_vit_layout* layout, void* out, hwocr_stream_t stream);

typedef struct {
  const void *in_norm_w, *qkv_w, *qkv_b, *o_w, *post_norm_w, *gate_up_w, *down_w;
  const void *qkv_wt, *o_wt, *gate_up_wt, *down_wt; /* fragment-tiled copies for decode (NULL: use the row-major ones) */
  hwocr_w8 qkv8, o8, gate_up8, down8; /* optional E4M3 copies for the prefill GEMMs (row-major codes + per-feature scales) */
  const void *qkv8t, *o8t, *gate_up8t, *down8t; /* the same codes byte-tiled for decode (hwocr_tile_weights_fp8; scales: above), or NULL */
} hwocr_dec_layer;

typedef struct {
  int layers, hidden, Hq, Hkv, inter, vocab, sec0, sec1;
  int head_dim; /* 128, or 256 (Gemma: KV cache in the row layout) */
  int gemma;    /* Gemma conventions (HF gemma/modeling_gemma.py): norms scale by (1 + w) in fp32, tanh-GELU gate, token
                 * embeddings times bf16(embed_scale), the prompt is a bidirectional prefix (paligemma/modeling_paligemma.py:256-262) */
  float eps, embed_scale;
  const void* embed;        /* [vocab][hidden] */
  const void* lm_head;      /* [vocab][hidden] (may alias embed) */
  const void* lm_head_t;    /* fragment-tiled copy of lm_head for decode (may be NULL) */
  hwocr_w8 lm_head8t;       /* E4M3 LM head for decode: byte-tiled codes (hwocr_tile_weights_fp8) + per-row scales, or {NULL, NULL} */
  const void* final_norm_w;
  const hwocr_dec_layer* L; /* host array[layers] */
  const void *rope_cos, *rope_sin; /* bf16 [max_pos][head_dim/2] */
  int max_pos;                     /* rows of the rope tables */
} hwocr_decoder;

typedef struct { /* KV cache: K [layer][seq][Hkv][ctx][128], VT [layer][seq][Hkv][128][ctx]; tiled != 0: every
                  * (seq, kv head) region is stored in the fragment-tiled order of csrc/common.h (kv_tiled_k/_v).
                  * fp8 != 0 (head_dim 256 only, the fp8 configuration: BASELINE config 4): k / vt hold E4M3 CODES, one byte per
                  * element, every (seq, kv head) region of ctx * 256 bytes in the operand order of csrc/common.h (kv8_k / kv8_v:
                  * 32-key blocks, a lane's 16 bytes = its 8 codes of two consecutive k-steps / d-tiles), with ONE fp32 scale per
                  * cached token and kv head: k_scale, v_scale [layer][seq][Hkv][ctx] (value = code * scale; scale = max|x| / 448
                  * over the token's 256 features, as hwocr_quant_rows_fp8).  Half the bytes a decode step streams. */
  void* k; void* vt; int nseq_max, ctx, tiled;
  float *k_scale, *v_scale; int fp8;
} hwocr_kv;

typedef struct {
  void *h, *hn, *qkv, *q, *attn, *act;   /* bf16 [rows][...]: hidden, hidden, (Hq + 2 Hkv) * head_dim, Hq * head_dim, Hq * head_dim, inter wide;
                                          * rows = nseq * rows_per_seq for hwocr_prefill, nseq for hwocr_decode_step (which does not touch qkv) */
  float *slabs;                           /* fp32 split-K slabs of a decode step: at least hwocr_decode_slab_floats(m, nseq) elements for every
                                          * nseq the workspace is used with (at <= 16 reads the QKV slab sits behind the down projection's) */
  float *part_o, *part_ml;                /* decode attention partials */
  int *arrive;                            /* [nseq_max][Hkv] arrival counters of hwocr_attn_decode, zero-initialised (or NULL: merge launch) */
  int *select_ws;                         /* [nseq_max][HWOCR_SELECT_WS_INTS] of hwocr_argmax_advance, zero-initialised (or NULL) */
  void *logits;                           /* bf16 [nseq][vocab] */
  void* q8;                               /* E4M3 staging of one prefill GEMM input, rows * max(hidden, Hq*head_dim, inter) bytes, or NULL */
  float* q8s;                             /* its row scales [rows] */
  void *kt, *vtt;                         /* hwocr_kv.fp8 only: bf16 K [nseq][Hkv][rows_per_seq][256] and V^T [nseq][Hkv][256][rows_per_seq] of ONE
                                           * prefill call and layer (the prefill attention reads them; hwocr_kv_quant_fp8 then fills the cache) */
} hwocr_dec_ws;

typedef struct {
  int *cur_ids, *lens, *n_gen, *finished, *out_tokens, *rope_delta; /* device int32 */
  int max_new, min_new, n_eos, pad_id;
  int eos[4];
  unsigned* seen; /* repetition-penalty bitmap [reads][seen_ld] (see hwocr_argmax_advance) or NULL */
  int seen_ld;
  float rep_penalty;
  int* status; /* device int32 (or NULL): HWOCR_STATUS_* bits raised by decode steps that met a read outside its invariants;
                * the host reads it whenever it synchronises and treats non-zero as an error */
  int do_sample; /* 0: greedy (hwocr_argmax_advance); 1: hwocr_sample_advance with the four fields below */
  float temperature;
  int top_k;
  float top_p;
  unsigned long long seed;
  const int* read_ids; /* device int32 [reads] (or NULL: the slot index): the caller's number of the read in each slot; it enters the
                        * RNG counter, so a read draws the same tokens whichever slot / batch it is decoded in */
} hwocr_gen_state;

/* prefill nseq reads laid out [nseq][rows_per_seq]; writes KV for reads seq0.. and the first generated token */
int hwocr_prefill(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv, const hwocr_gen_state* st,
                  const int* ids, const int* img_row, const void* img_embeds, const int* pos3, const int* seq_lens,
                  const int* last_rows, int nseq, int rows_per_seq, int seq0, int max_len, hwocr_stream_t stream);

/* one greedy token for every read in flight (reads 0..nseq-1 of the cache) */
int hwocr_decode_step(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv, const hwocr_gen_state* st,
                      int nseq, int attn_splits, hwocr_stream_t stream);

/* fp32 elements hwocr_decode_step writes into ws->slabs at nseq reads in flight for this decoder (split-K slabs of the QKV / o / down
 * projections; at <= 16 reads the down projection's slabs AND the QKV slab behind them).  Host only, touches no device memory;
 * -1 on bad arguments.  A workspace must hold the maximum over every nseq it is used with. */
long hwocr_decode_slab_floats(const hwocr_decoder* m, int nseq);

/* Plan recording: between hwocr_plan_begin() and hwocr_plan_end() every launcher of this library called on the SAME thread checks
 * its arguments as usual, records one text line "<kernel instance> <geometry>" and returns HWOCR_OK without touching the device
 * (pointers are never dereferenced on the host either, apart from the model / layout structs).  hwocr_vit_forward / hwocr_prefill /
 * hwocr_decode_step under it list the launches of a configuration: tests/test_wide_variants.py holds that list against the parity
 * cases.  hwocr_plan_end copies the '\n'-separated lines (NUL-terminated) into buf and stops recording; *needed = bytes required
 * (HWOCR_EINVAL if len is smaller: call it again with a larger buffer). */
int hwocr_plan_begin(void);
int hwocr_plan_end(char* buf, int len, int* needed);

/* Partitioning the chip between streams (pipeline.LanePipeline, order "partition"): a stream whose queue may use only the CUs whose
 * bit is set in mask[words] (hipExtStreamCreateWithCUMask; consecutive bits are dealt round-robin over the XCDs by the driver), the
 * number of CUs the calling thread's persistent launches should size their grids to (0 = whole device), and a probe that reports
 * where n_wg one-wave workgroups ran: out[n_wg][2] = (XCC_ID, HW_ID) hardware registers. */
int hwocr_stream_create_cumask(const unsigned int* mask, int words, void** stream_out);
int hwocr_stream_destroy(void* stream);
int hwocr_set_cu_budget(int cus);
int hwocr_probe_placement(unsigned int* out, int n_wg, long spin_cycles, hwocr_stream_t stream);

/* capture one decode step into a HIP graph; replay it n times back-to-back on `stream` */
int hwocr_decode_graph_create(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv,
                              const hwocr_gen_state* st, int nseq, int attn_splits, void** graph_out);
int hwocr_decode_graph_launch(void* graph, int n, hwocr_stream_t stream);
int hwocr_decode_graph_destroy(void* graph);

/* ---- strategy preprocessing on the device (csrc/imagepre.hip): the reference's transforms in their PIL-fallback form
 * (ocr_agent/tools.py:514-516 high_contrast, :530-531 binarize, :544-546 sharpen) and the image processor's bicubic resize
 * (HF image_processing_pil_qwen2_vl.py:152-183 -> Pillow Resample.c), bit-identical to Pillow.  Images: uint8 [H][W][3]. */
int hwocr_img_luma_sum(const void* rgb, long npix, unsigned long long* sum, hwocr_stream_t stream); /* sum of convert("L") */
int hwocr_img_contrast(const void* src, void* dst, long nbytes, int mean, float factor, hwocr_stream_t stream);
/* the same with mean = int(*sum / npix + 0.5) computed on the device from hwocr_img_luma_sum's result (same stream): ImageEnhance.Contrast
 * without a host round trip between the two kernels */
int hwocr_img_contrast_dev(const void* src, void* dst, long nbytes, const unsigned long long* sum, long npix, float factor,
                           hwocr_stream_t stream);
int hwocr_img_binarize(const void* rgb, void* dst_rgb, long npix, hwocr_stream_t stream);
int hwocr_img_sharpen(const void* src, void* dst, int H, int W, hwocr_stream_t stream); /* src != dst */
/* tmp: H * out_w * 3 bytes; bounds [n_out][2] = (first tap, taps), coef [n_out][ksize] = Pillow's 22-bit fixed-point taps */
int hwocr_img_resize_bicubic(const void* src, void* tmp, void* dst, int H, int W, int out_h, int out_w, const int* h_bounds,
                             const int* h_coef, int h_ksize, const int* v_bounds, const int* v_coef, int v_ksize,
                             hwocr_stream_t stream);

/* bench instrumentation: HIP events on the launch stream around every hwocr_gemm_wide (on == 1) or hwocr_gemm_wide_fp8
 * (on == 2) launch while enabled */
int hwocr_profile_enable(int on);
int hwocr_profile_read(double* total_ms, double* total_flops, long* launches);

/* ---- native text kernels (libhwocr_text.so, host C++) ------------------------------------------------------- */

/* unit-cost edit distance over code points / word ids; replaces tools.py:69-100 */
int64_t hwocr_levenshtein_u32(const uint32_t* a, int64_t n, const uint32_t* b, int64_t m);
/* LCS alignment of `words` to `backbone` over case-folded word ids; out[i] = index into words or -1.
 * Replaces tools.py:465-493 (same tie-breaking in the backtrack). */
int hwocr_lcs_align_u32(const uint32_t* backbone, int64_t n, const uint32_t* words, int64_t m, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
