// Model-level launch sequences: what `model.generate` (ocr_agent/tools.py:764-765) expands to for the
// Qwen2-VL family, expressed as kernel launches on one HIP stream with caller-owned buffers.
//   hwocr_vit_forward  <- Qwen2VisionTransformerPretrainedModel.forward (HF modeling_qwen2_vl.py:700-729) and
//                         Qwen2_5_VisionTransformerPretrainedModel.forward (HF modeling_qwen2_5_vl.py:407-481)
//   hwocr_prefill      <- Qwen2VLModel.forward splice + Qwen2VLTextModel.forward + lm_head (… :790-872, :1383-1387)
//   hwocr_decode_step  <- one iteration of GenerationMixin._sample's while loop (HF generation/utils.py:2876-2941)
// The decode step touches only device state (token ids, lengths, stop flags live in HBM), so it can be captured
// once into a HIP graph and replayed for every generated token without a host round trip.
#include "common.h"
#include "hwocr.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(call)                \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != HWOCR_OK) return rc_; \
  } while (0)

namespace {
inline bf16* B(void* p) { return (bf16*)p; }
inline const bf16* B(const void* p) { return (const bf16*)p; }
inline bool hwocr_vit_qkv_fusable(int M, int heads, int hd) {  // == gemm::vit_qkv_fusable (gemm_common.h)
  return M >= 1024 && (heads * hd) % 256 == 0 && hd % 16 == 0 && hd <= 128;
}
inline int pick_splitk(int K, int N, int want_wgs) {
  const int chunks = (K + 255) / 256;
  const int tiles = (N + 31) / 32;  // gemm_skinny's small-N configuration: 32 weight rows per workgroup
  int s = (want_wgs + tiles - 1) / tiles;
  if (s < 1) s = 1;
  if (s > chunks) s = chunks;
  // every slice must own at least one chunk
  const int per = (chunks + s - 1) / s;
  s = (chunks + per - 1) / per;
  return s;
}
// gemm_stream (<= 128 reads, tiled weights) fills the chip by itself; K is split to cut the activation re-staging
// (x bytes into the CUs = 64 KiB x K / splitk against 1 KiB x N x splitk of slab): measured best 8 slabs, 12 for the
// long-K / narrow-N down projection, and slices of at least 6 K tiles (K = 1536: 4 slabs; GEMM + the slab-summing consumer,
// tools/bench_decode_gemm.py sweep at 126 and 252 reads)
// Above 128 reads the 64-wide-K kernel wants at most 8 weight tiles per workgroup (N/16 tiles over 256/s groups): the split is
// lowered until that holds (7B down projection: 12 -> 9 slabs, 64 -> 56 us with its slab-summing consumer).
inline int pick_splitk_stream(int K, int N, int rows) {
  const int ktiles = K / 64;
  // above 128 reads (two row blocks per weight-tile group, gemm_stream.hip) fewer, longer slices win: sweep at 252 reads,
  // GEMM + slab-summing consumer: 2B down 8-10 slabs 21.3 us (12: 22.7), 7B o 4 slabs 20.2 (8: 26.2), 7B qkv 4: 17.2 (7: 18.6)
  int s = K >= 4 * N ? (rows > 128 ? 8 : 12) : (rows > 128 ? 4 : 8);
  if (s > ktiles / 6) s = ktiles / 6;
  if (rows > 128)
    while (s > 1 && (N / 16) * s > 8 * 256) --s;
  if (s < 1) s = 1;
  const int per = (ktiles + s - 1) / s;  // every slice must own at least one K tile
  return (ktiles + per - 1) / per;
}
// One wide GEMM of the read path (ldx == ldw == K everywhere).  With an E4M3 copy of the weight and K a multiple of 128
// the activation rows are quantised into the workspace and the product runs on the fp8 MFMA; else bf16.
inline bool runs_fp8(const hwocr_w8& w8, const void* q8, const float* q8s, int K) {
  return w8.w && w8.scale && q8 && q8s && (K % 128) == 0;
}
// quantised: the producer (a norm) already left the E4M3 rows + scales of X in the workspace
inline int wide(const void* X, const void* W, const hwocr_w8& w8, void* q8, float* q8s, const void* bias, const void* res,
                void* out, int M, int N, int K, int ldo, int ldres, int epi, hipStream_t st, bool quantised = false) {
  if (runs_fp8(w8, q8, q8s, K)) {
    if (!quantised) CHECK(hwocr_quant_rows_fp8(X, q8, q8s, M, K, K, K, st));
    return hwocr_gemm_wide_fp8(q8, q8s, w8.w, w8.scale, bias, res, out, M, N, K, K, K, ldo, ldres, epi, st);
  }
  return hwocr_gemm_wide(X, W, bias, res, out, M, N, K, K, K, ldo, ldres, epi, st);
}
}  // namespace

extern "C" int hwocr_abi_version(void) { return HWOCR_ABI_VERSION; }

// ---- plan recording (common.h): one text line per launch the calling thread would have made
#include <cstdarg>
#include <cstring>
#include <string>
static thread_local bool t_plan_on = false;
static thread_local std::string t_plan;
bool hwocr_plan_on() { return t_plan_on; }
void hwocr_plan_note(const char* fmt, ...) {
  char line[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(line, sizeof(line), fmt, ap);
  va_end(ap);
  t_plan += line;
  t_plan += '\n';
}
extern "C" int hwocr_plan_begin(void) {
  t_plan.clear();
  t_plan_on = true;
  return HWOCR_OK;
}
// copies the recorded lines ('\n'-separated, NUL-terminated) into buf; *needed = bytes required (call again with a larger
// buffer if it exceeds len).  Recording stops either way.
extern "C" int hwocr_plan_end(char* buf, int len, int* needed) {
  t_plan_on = false;
  if (needed) *needed = (int)t_plan.size() + 1;
  if (!buf || len < (int)t_plan.size() + 1) return HWOCR_EINVAL;
  memcpy(buf, t_plan.c_str(), t_plan.size() + 1);
  return HWOCR_OK;
}

// ---- partitioning the chip between streams (pipeline.py: one batch's tower + prefill beside another batch's decode)
// A stream whose queue may only use the CUs of `mask` (bit i = CU i in the driver's enumeration; on gfx942 / gfx950 the kernel driver
// deals consecutive bits round-robin over the 8 XCDs, so a prefix of n bits is n / 8 CUs of EVERY XCD and the XCD-aware tile mappings
// keep their meaning - hwocr_probe_placement shows what a mask really gives).
extern "C" int hwocr_stream_create_cumask(const unsigned int* mask, int words, void** stream_out) {
  if (!mask || words <= 0 || !stream_out) return HWOCR_EINVAL;
  (void)hipGetLastError();
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return hwocr_launch_status();
  *stream_out = s;
  return HWOCR_OK;
}
extern "C" int hwocr_stream_destroy(void* stream) {
  if (!stream) return HWOCR_EINVAL;
  (void)hipGetLastError();
  if (hipStreamDestroy((hipStream_t)stream) != hipSuccess) return hwocr_launch_status();
  return HWOCR_OK;
}
// The CUs the calling thread's launches may count on (0 = the whole device): persistent kernels size their grid to it, so that a
// launch into a CU-masked stream is one round of workgroups and not one and a third.
static thread_local int t_cu_budget = 0;
int hwocr_cu_budget() { return t_cu_budget; }
extern "C" int hwocr_set_cu_budget(int cus) {
  if (cus < 0) return HWOCR_EINVAL;
  t_cu_budget = cus;
  return HWOCR_OK;
}
namespace {
__global__ __launch_bounds__(64) void probe_placement_kernel(unsigned int* out, long spin) {
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: cu [11:8], sh [12], se [15:13]
  }
}
}  // namespace
// out[n_wg][2] = (XCC_ID, HW_ID) of the CU each of n_wg one-wave workgroups ran on; each holds its CU for spin_cycles
extern "C" int hwocr_probe_placement(unsigned int* out, int n_wg, long spin_cycles, hipStream_t st) {
  if (!out || n_wg <= 0 || spin_cycles < 0) return HWOCR_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(probe_placement_kernel, dim3(n_wg), dim3(64), 0, st, out, spin_cycles);
  return hwocr_launch_status();
}

static char g_last_error[256] = "";
extern "C" void hwocr_record_error(const char* where, int hip_error, const char* text) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: HIP error %d (%s)", where, hip_error, text ? text : "?");
}
extern "C" const char* hwocr_last_error(void) { return g_last_error; }

extern "C" int hwocr_vit_forward(const hwocr_vit* m, const hwocr_vit_ws* ws, const void* images, int nimg, int H,
                                 int W, int rows_per_img_ld, const hwocr_vit_layout* lay, void* out, hipStream_t st) {
  if (!m || !ws || !lay || nimg <= 0 || rows_per_img_ld % 64) return HWOCR_EINVAL;
  const int D = m->dim, rows = nimg * rows_per_img_ld;
  const int hd = m->head_pad ? m->head_pad : D / m->heads;  // width of a head in Q / K / V^T (zero-padded for SigLIP)
  const int DH = m->heads * hd;                              // attention width: qkv_w is [3 DH][D], proj_w [D][DH]
  const int P = (H / m->patch) * (W / m->patch);
  const int mm = m->merge * m->merge;
  const bool v25 = m->kind == HWOCR_VIT_QWEN2_5, sig = m->kind == HWOCR_VIT_SIGLIP;
  // pre-attention / pre-MLP / merger norm of the family
  auto norm = [&](const void* w, const void* b) {
    return v25 ? hwocr_add_rmsnorm(nullptr, 0, 0, 0, nullptr, ws->x, D, w, ws->xn, D, nullptr, rows, D, m->eps, 0, st)
               : hwocr_layernorm(ws->x, w, b, ws->xn, rows, D, D, D, m->eps, st);
  };
  // the same norm in front of a GEMM that runs in fp8: E4M3 rows + scales straight into the workspace
  auto norm8 = [&](const void* w, const void* b) {
    return v25 ? hwocr_rmsnorm_fp8(ws->x, D, w, ws->q8, ws->q8s, D, rows, D, m->eps, 0, st)
               : hwocr_layernorm_fp8(ws->x, w, b, ws->q8, ws->q8s, rows, D, D, D, m->eps, st);
  };
  CHECK(hwocr_patchify(images, m->pixel_lut, ws->patches, nimg, H, W, m->patch, m->merge, m->tps, m->kpad,
                       rows_per_img_ld, lay->row_src, st));
  if (sig) {
    // bf16(bf16(conv + bias) + position): the learned positions ride in as the residual operand, page by page
    if (!m->pos_embed) return HWOCR_EINVAL;
    for (int i = 0; i < nimg; ++i) {
      const long r0 = (long)i * rows_per_img_ld;
      CHECK(hwocr_gemm_wide(B(ws->patches) + r0 * m->kpad, m->patch_w, m->patch_b, m->pos_embed, B(ws->x) + r0 * D, P, D,
                            m->kpad, m->kpad, m->kpad, D, D, HWOCR_EPI_RESIDUAL, st));
    }
  } else {
    CHECK(hwocr_gemm_wide(ws->patches, m->patch_w, nullptr, nullptr, ws->x, rows, D, m->kpad, m->kpad, m->kpad, D, 0,
                          HWOCR_EPI_LINEAR, st));
  }
  const float scale = 1.0f / sqrtf((float)(D / m->heads));
  static const bool fuse_env = HWOCR_DIAG_ENV_INT("HWOCR_VIT_FUSE_QKV", 1) != 0;
  const bool fuse_qkv = fuse_env && m->qk_interleaved && (rows % 64) == 0 && hwocr_vit_qkv_fusable(rows, m->heads, hd);
  for (int l = 0; l < m->depth; ++l) {
    const hwocr_vit_block& b = m->blocks[l];
    const bool q1 = runs_fp8(b.qkv8, ws->q8, ws->q8s, D), q2 = runs_fp8(b.fc18, ws->q8, ws->q8s, D);
    CHECK(q1 ? norm8(b.ln1_w, b.ln1_b) : norm(b.ln1_w, b.ln1_b));
    if (fuse_qkv) {  // rotary + head split + V transpose in the GEMM's epilogue: the [rows][3 DH] intermediate never exists
      const hwocr_vit_split sp{ws->q, ws->k, ws->vt, lay->pos_h, lay->pos_w, m->rope_cos, m->rope_sin, m->heads, hd, rows};
      if (q1)
        CHECK(hwocr_gemm_vit_qkv(ws->q8, b.qkv8.w, b.qkv_b, rows, D, D, D, ws->q8s, b.qkv8.scale, &sp, st));
      else
        CHECK(hwocr_gemm_vit_qkv(ws->xn, b.qkv_w, b.qkv_b, rows, D, D, D, nullptr, nullptr, &sp, st));
    } else {
      CHECK(wide(ws->xn, b.qkv_w, b.qkv8, ws->q8, ws->q8s, b.qkv_b, nullptr, ws->qkv, rows, 3 * DH, D, 3 * DH, 0,
                 HWOCR_EPI_LINEAR, st, q1));
      CHECK(hwocr_vit_rope_split(ws->qkv, ws->q, ws->k, ws->vt, lay->pos_h, lay->pos_w, m->rope_cos, m->rope_sin, rows,
                                 rows, m->heads, hd, m->qk_interleaved, st));
    }
    if (v25 && b.windowed && lay->nwin > 0) {
      CHECK(hwocr_attn_varlen(ws->q, ws->k, ws->vt, ws->attn, lay->win_off, lay->win_lens, lay->nwin, m->heads, hd,
                              lay->max_win, (long)rows * hd, hd, (long)rows * hd, hd, (long)hd * rows, rows, DH, scale,
                              st));
    } else {
      CHECK(hwocr_attn_prefill(ws->q, ws->k, ws->vt, ws->attn, lay->seg_lens, nimg, m->heads, 1, hd, P, 0,
                               (long)rows_per_img_ld * hd, (long)rows * hd, hd,   // Q [head][rows][hd]
                               (long)rows_per_img_ld * hd, (long)rows * hd, hd,   // K
                               rows_per_img_ld, (long)hd * rows, rows,            // V^T [head][hd][rows]
                               (long)rows_per_img_ld * DH, DH, scale, 0, st));
    }
    CHECK(wide(ws->attn, b.proj_w, b.proj8, ws->q8, ws->q8s, b.proj_b, ws->x, ws->x, rows, D, DH, D, D, HWOCR_EPI_RESIDUAL,
               st));
    CHECK(q2 ? norm8(b.ln2_w, b.ln2_b) : norm(b.ln2_w, b.ln2_b));
    if (v25) {  // down(silu(gate(x)) * up(x)), all three with bias (HF modeling_qwen2_5_vl.py:84-96)
      CHECK(wide(ws->xn, b.fc1_w, b.fc18, ws->q8, ws->q8s, b.fc1_b, nullptr, ws->mlp, rows, 2 * m->mlp_dim, D, m->mlp_dim, 0,
                 HWOCR_EPI_SWIGLU, st, q2));
    } else {
      CHECK(wide(ws->xn, b.fc1_w, b.fc18, ws->q8, ws->q8s, b.fc1_b, nullptr, ws->mlp, rows, m->mlp_dim, D, m->mlp_dim, 0,
                 sig ? HWOCR_EPI_GELU_TANH : HWOCR_EPI_QUICKGELU, st, q2));
    }
    CHECK(wide(ws->mlp, b.fc2_w, b.fc28, ws->q8, ws->q8s, b.fc2_b, ws->x, ws->x, rows, D, m->mlp_dim, D, D,
               HWOCR_EPI_RESIDUAL, st));
  }
  // patch merger: norm -> view(-1, merge^2 * D) -> Linear -> GELU -> Linear; SigLIP: post_layernorm -> one projector
  CHECK(norm(m->merger_ln_w, m->merger_ln_b));
  if (sig)
    return hwocr_gemm_wide(ws->xn, m->merger_fc2_w, m->merger_fc2_b, nullptr, out, rows, m->out_dim, D, D, D, m->out_dim, 0,
                           HWOCR_EPI_LINEAR, st);
  const int mrows = rows / mm, MD = mm * D;
  CHECK(hwocr_gemm_wide(ws->xn, m->merger_fc1_w, m->merger_fc1_b, nullptr, ws->merge_mid, mrows, MD, MD, MD, MD, MD, 0,
                        HWOCR_EPI_GELU, st));
  CHECK(hwocr_gemm_wide(ws->merge_mid, m->merger_fc2_w, m->merger_fc2_b, nullptr, out, mrows, m->out_dim, MD, MD, MD,
                        m->out_dim, 0, HWOCR_EPI_LINEAR, st));
  return HWOCR_OK;
}

// one decode GEMM: E4M3 weights when the layer carries a byte-tiled copy, else the fragment-tiled bf16 copy, else row-major
static int decode_gemm(const void* x, const void* w, const void* wt, const void* w8t, const float* w8s, void* out, int nseq, int N,
                       int K, int ldo, int epi, int splitk, hipStream_t st) {
  if (w8t && w8s) return hwocr_gemm_skinny_w8(x, w8t, w8s, nullptr, out, nseq, N, K, K, ldo, epi, splitk, st);
  return hwocr_gemm_skinny(x, wt ? wt : w, nullptr, out, nseq, N, K, K, K, ldo, epi, splitk, wt != nullptr, st);
}

extern "C" int hwocr_prefill(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv,
                             const hwocr_gen_state* gs, const int* ids, const int* img_row, const void* img_embeds,
                             const int* pos3, const int* seq_lens, const int* last_rows, int nseq, int rows_per_seq,
                             int seq0, int max_len, hipStream_t st) {
  if (!m || !ws || !kv || !gs || nseq <= 0 || rows_per_seq % 64 || rows_per_seq > kv->ctx ||
      seq0 + nseq > kv->nseq_max || nseq > 128 || (m->head_dim != 128 && m->head_dim != 256) ||
      (m->head_dim != 128 && kv->tiled))
    return HWOCR_EINVAL;
  // E4M3 cache (hwocr_kv.fp8): this call's K / V^T stay bf16 in the workspace (ws->kt / ws->vtt: the prefill attention reads them),
  // then hwocr_kv_quant_fp8 fills the cache
  if (kv->fp8 && (m->head_dim != 256 || kv->tiled || !kv->k_scale || !kv->v_scale || !ws->kt || !ws->vtt || (kv->ctx % 32))) return HWOCR_EINVAL;
  const int HD = m->head_dim, G = m->gemma;
  const int rows = nseq * rows_per_seq, Hd = m->hidden, QW = (m->Hq + 2 * m->Hkv) * HD;
  const long k_head = (long)kv->ctx * HD, k_seq = (long)m->Hkv * k_head, k_layer = (long)kv->nseq_max * k_seq;
  const float scale = 1.0f / sqrtf((float)HD);
  CHECK(hwocr_embed_splice(ids, img_row, m->embed, img_embeds, ws->h, rows, Hd, G ? m->embed_scale : 1.0f, st));
  for (int l = 0; l < m->layers; ++l) {
    const hwocr_dec_layer& L = m->L[l];
    bf16* Kc = kv->fp8 ? B(ws->kt) : B(kv->k) + l * k_layer + seq0 * k_seq;
    bf16* Vc = kv->fp8 ? B(ws->vtt) : B(kv->vt) + l * k_layer + seq0 * k_seq;
    // strides of where this layer's K / V^T live while the prompt is attended over: the cache, or the per-call bf16 scratch
    const long a_head = kv->fp8 ? (long)rows_per_seq * HD : k_head, a_seq = kv->fp8 ? (long)m->Hkv * a_head : k_seq;
    const int a_ctx = kv->fp8 ? rows_per_seq : kv->ctx;
    const bool q1 = runs_fp8(L.qkv8, ws->q8, ws->q8s, Hd), q2 = runs_fp8(L.gate_up8, ws->q8, ws->q8s, Hd);
    if (q1)
      CHECK(hwocr_rmsnorm_fp8(ws->h, Hd, L.in_norm_w, ws->q8, ws->q8s, Hd, rows, Hd, m->eps, G, st));
    else
      CHECK(hwocr_add_rmsnorm(nullptr, 0, 0, 0, nullptr, ws->h, Hd, L.in_norm_w, ws->hn, Hd, nullptr, rows, Hd, m->eps,
                              G, st));
    CHECK(wide(ws->hn, L.qkv_w, L.qkv8, ws->q8, ws->q8s, L.qkv_b, nullptr, ws->qkv, rows, QW, Hd, QW, 0, HWOCR_EPI_LINEAR,
               st, q1));
    CHECK(hwocr_mrope_kv_prefill(ws->qkv, ws->q, Kc, Vc, pos3, m->rope_cos, m->rope_sin, rows, rows_per_seq, m->Hq,
                                 m->Hkv, m->sec0, m->sec1, a_seq, a_head, a_seq, a_head, a_ctx, HD, kv->tiled, st));
    // Gemma (PaliGemma): the whole prompt is a bidirectional prefix; Qwen: causal
    CHECK(hwocr_attn_prefill(ws->q, Kc, Vc, ws->attn, seq_lens, nseq, m->Hq, m->Hq / m->Hkv, HD, max_len, G ? 0 : 1,
                             (long)rows_per_seq * m->Hq * HD, HD, (long)m->Hq * HD,  // Q [row][Hq][128]
                             a_seq, a_head, HD, a_seq, a_head, a_ctx,
                             (long)rows_per_seq * m->Hq * HD, (long)m->Hq * HD, scale, kv->tiled, st));
    if (kv->fp8) {  // the cache's share of this layer: codes + one scale per token, for every position of the padded prompt
      const long s_layer = (long)kv->nseq_max * m->Hkv * kv->ctx, s_seq = (long)m->Hkv * kv->ctx;
      CHECK(hwocr_kv_quant_fp8(Kc, Vc, a_seq, a_head, a_seq, a_head, a_ctx, (unsigned char*)kv->k + l * k_layer + seq0 * k_seq,
                               (unsigned char*)kv->vt + l * k_layer + seq0 * k_seq, kv->k_scale + l * s_layer + seq0 * s_seq,
                               kv->v_scale + l * s_layer + seq0 * s_seq, nseq, m->Hkv, rows_per_seq, kv->ctx, st));
    }
    CHECK(wide(ws->attn, L.o_w, L.o8, ws->q8, ws->q8s, nullptr, ws->h, ws->h, rows, Hd, m->Hq * HD, Hd, Hd,
               HWOCR_EPI_RESIDUAL, st));
    if (q2)
      CHECK(hwocr_rmsnorm_fp8(ws->h, Hd, L.post_norm_w, ws->q8, ws->q8s, Hd, rows, Hd, m->eps, G, st));
    else
      CHECK(hwocr_add_rmsnorm(nullptr, 0, 0, 0, nullptr, ws->h, Hd, L.post_norm_w, ws->hn, Hd, nullptr, rows, Hd,
                              m->eps, G, st));
    CHECK(wide(ws->hn, L.gate_up_w, L.gate_up8, ws->q8, ws->q8s, nullptr, nullptr, ws->act, rows, 2 * m->inter, Hd, m->inter,
               0, G ? HWOCR_EPI_GEGLU : HWOCR_EPI_SWIGLU, st, q2));
    CHECK(wide(ws->act, L.down_w, L.down8, ws->q8, ws->q8s, nullptr, ws->h, ws->h, rows, Hd, m->inter, Hd, Hd,
               HWOCR_EPI_RESIDUAL, st));
  }
  // final norm on the last prompt token of every read -> LM head -> first generated token
  CHECK(hwocr_add_rmsnorm(nullptr, 0, 0, 0, nullptr, ws->h, Hd, m->final_norm_w, ws->hn, Hd, last_rows, nseq, Hd,
                          m->eps, G, st));
  CHECK(decode_gemm(ws->hn, m->lm_head, m->lm_head_t, m->lm_head8t.w, m->lm_head8t.scale, ws->logits, nseq, m->vocab, Hd, m->vocab,
                    HWOCR_EPI_LINEAR, 1, st));  // the same LM head (bf16 or E4M3) as every later step
  if (gs->do_sample) {
    if (!gs->read_ids) return HWOCR_EINVAL;  // (a chunk's rows would otherwise be numbered from 0: not the decode step's numbers)
    CHECK(hwocr_sample_advance(ws->logits, m->vocab, m->vocab, nseq, gs->cur_ids + seq0, gs->lens + seq0, gs->n_gen + seq0,
                               gs->finished + seq0, gs->out_tokens + (long)seq0 * gs->max_new, gs->max_new, gs->min_new, gs->eos,
                               gs->n_eos, gs->pad_id, gs->seen ? gs->seen + (long)seq0 * gs->seen_ld : nullptr, gs->seen_ld,
                               gs->rep_penalty, gs->temperature, gs->top_k, gs->top_p, gs->seed, gs->read_ids + seq0, nullptr, st));
    return HWOCR_OK;
  }
  CHECK(hwocr_argmax_advance(ws->logits, m->vocab, m->vocab, nseq, gs->cur_ids + seq0, gs->lens + seq0,
                             gs->n_gen + seq0, gs->finished + seq0, gs->out_tokens + (long)seq0 * gs->max_new,
                             gs->max_new, gs->min_new, gs->eos, gs->n_eos, gs->pad_id,
                             gs->seen ? gs->seen + (long)seq0 * gs->seen_ld : nullptr, gs->seen_ld, gs->rep_penalty,
                             ws->select_ws, st));  // (a prefill chunk's rows use the first nseq counters: the stream orders the chunks)
  return HWOCR_OK;
}


// this layer's attention of a decode step (fused q / k / v finish) over the bf16 cache or, hwocr_kv.fp8, over the E4M3 one
static int decode_attention(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv, const hwocr_gen_state* gs, int l,
                            const float* slabs, int nslab, const void* bias, int* arrive, int nseq, int attn_splits, hipStream_t st) {
  const int HD = m->head_dim, QW = (m->Hq + 2 * m->Hkv) * HD;
  const long k_head = (long)kv->ctx * HD, k_seq = (long)m->Hkv * k_head, k_layer = (long)kv->nseq_max * k_seq;
  const float scale = 1.0f / sqrtf((float)HD);
  if (kv->fp8) {
    const long s_layer = (long)kv->nseq_max * m->Hkv * kv->ctx;
    return hwocr_attn_decode_qkv_fp8kv(slabs, nslab, (long)nseq * QW, bias, (unsigned char*)kv->k + l * k_layer, (unsigned char*)kv->vt + l * k_layer,
                                       kv->k_scale + l * s_layer, kv->v_scale + l * s_layer, gs->lens, gs->rope_delta, m->rope_cos, m->rope_sin,
                                       ws->attn, ws->part_o, ws->part_ml, arrive, nseq, m->Hq, m->Hkv, attn_splits, scale, kv->ctx, m->max_pos,
                                       gs->status, st);
  }
  return hwocr_attn_decode_qkv(slabs, nslab, (long)nseq * QW, bias, B(kv->k) + l * k_layer, B(kv->vt) + l * k_layer, gs->lens, gs->rope_delta,
                               m->rope_cos, m->rope_sin, ws->attn, ws->part_o, ws->part_ml, arrive, nseq, m->Hq, m->Hkv, attn_splits, k_seq,
                               k_head, k_seq, k_head, kv->ctx, scale, HD, kv->tiled, kv->ctx, m->max_pos, gs->status, st);
}

// split-K of the three slab-producing GEMMs of a decode step (qkv, o, down) at `nseq` reads in flight
static void decode_splits(const hwocr_decoder* m, int nseq, int& s_qkv, int& s_o, int& s_d) {
  const int HD = m->head_dim, Hd = m->hidden, QW = (m->Hq + 2 * m->Hkv) * HD, OW = m->Hq * HD;
  bool stream = (Hd % 64) == 0 && (OW % 64) == 0 && (m->inter % 64) == 0;
  for (int l = 0; l < m->layers && stream; ++l)
    stream = (m->L[l].qkv_wt || m->L[l].qkv8t) && (m->L[l].o_wt || m->L[l].o8t) && (m->L[l].down_wt || m->L[l].down8t);
  s_qkv = stream ? pick_splitk_stream(Hd, QW, nseq) : pick_splitk(Hd, QW, 400);
  s_o = stream ? pick_splitk_stream(OW, Hd, nseq) : pick_splitk(OW, Hd, 400);
  s_d = stream ? pick_splitk_stream(m->inter, Hd, nseq) : pick_splitk(m->inter, Hd, 400);
}

// HWOCR_DECODE_LASTWG=0: split attention partials are merged, and the token picked, by launches of their own (A/B runs)
static bool decode_lastwg() {
  static const bool on = HWOCR_DIAG_ENV_INT("HWOCR_DECODE_LASTWG", 1) != 0;
  return on;
}

// the token selection that ends a decode step: logits -> next token, stop flags, bookkeeping (all on the device)
static int decode_select(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_gen_state* gs, int nseq, hipStream_t st) {
  if (gs->do_sample) {
    if (!gs->read_ids) return HWOCR_EINVAL;
    return hwocr_sample_advance(ws->logits, m->vocab, m->vocab, nseq, gs->cur_ids, gs->lens, gs->n_gen, gs->finished, gs->out_tokens,
                                gs->max_new, gs->min_new, gs->eos, gs->n_eos, gs->pad_id, gs->seen, gs->seen_ld, gs->rep_penalty,
                                gs->temperature, gs->top_k, gs->top_p, gs->seed, gs->read_ids, nullptr, st);
  }
  return hwocr_argmax_advance(ws->logits, m->vocab, m->vocab, nseq, gs->cur_ids, gs->lens, gs->n_gen, gs->finished,
                              gs->out_tokens, gs->max_new, gs->min_new, gs->eos, gs->n_eos, gs->pad_id, gs->seen, gs->seen_ld,
                              gs->rep_penalty, decode_lastwg() ? ws->select_ws : nullptr, st);
}

// <= 16 reads in flight and every layer GEMM with its bf16 fragment-tiled copy (no E4M3 decode weights): the 5-launch layer of
// csrc/gemm_rows16.hip.  HWOCR_DECODE_ROWS16=0: the general path at every read count (A/B runs).
static bool decode_takes_rows16(const hwocr_decoder* m, int nseq) {
  static const bool on = HWOCR_DIAG_ENV_INT("HWOCR_DECODE_ROWS16", 1) != 0;
  if (!on || nseq > 16 || m->hidden > 4096 || (m->hidden % 32) || (m->inter % 32) || ((m->Hq * m->head_dim) % 32)) return false;
  for (int l = 0; l < m->layers; ++l) {
    const hwocr_dec_layer& L = m->L[l];
    if (!L.qkv_wt || !L.o_wt || !L.gate_up_wt || !L.down_wt || L.qkv8t || L.o8t || L.gate_up8t || L.down8t) return false;
  }
  return true;
}
// split-K of the down projection on that path: enough slices to put a workgroup on most CUs (N / 16 tiles each), at most 4 (the
// consumer's norm prologue loads all slabs at once), slices of at least 16 k-steps
static int rows16_down_split(const hwocr_decoder* m) {
  const int tiles = m->hidden / 16, ksteps = m->inter / 32;
  int s = 256 / tiles;
  if (s > 4) s = 4;
  if (s > ksteps / 16) s = ksteps / 16;
  if (s < 1) s = 1;
  const int per = (ksteps + s - 1) / s;  // every slice must own at least one k-step
  return (ksteps + per - 1) / per;
}

// fp32 elements hwocr_decode_step writes into ws->slabs at `nseq` reads in flight (hwocr.h): the caller's size contract
extern "C" long hwocr_decode_slab_floats(const hwocr_decoder* m, int nseq) {
  if (!m || nseq <= 0 || nseq > 256 || !m->L) return -1;
  const long HD = m->head_dim, Hd = m->hidden, QW = (m->Hq + 2 * m->Hkv) * HD;
  if (decode_takes_rows16(m, nseq)) return ((long)rows16_down_split(m) * Hd + QW) * nseq;
  int s_qkv, s_o, s_d;
  decode_splits(m, nseq, s_qkv, s_o, s_d);
  long need = s_qkv * QW;
  if (s_o * Hd > need) need = s_o * Hd;
  if (s_d * Hd > need) need = s_d * Hd;
  return need * nseq;
}

extern "C" int hwocr_decode_step(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv,
                                 const hwocr_gen_state* gs, int nseq, int attn_splits, hipStream_t st) {
  if (!m || !ws || !kv || !gs || nseq <= 0 || nseq > 256 || nseq > kv->nseq_max || attn_splits < 1 || attn_splits > 16 ||
      m->max_pos < 1 || (m->head_dim != 128 && m->head_dim != 256) || (m->head_dim != 128 && kv->tiled))
    return HWOCR_EINVAL;
  if (kv->fp8 && (m->head_dim != 256 || kv->tiled || !kv->k_scale || !kv->v_scale || (kv->ctx % 32))) return HWOCR_EINVAL;
  const int HD = m->head_dim, G = m->gemma;
  const int Hd = m->hidden, QW = (m->Hq + 2 * m->Hkv) * HD, OW = m->Hq * HD;
  const long k_head = (long)kv->ctx * HD, k_seq = (long)m->Hkv * k_head, k_layer = (long)kv->nseq_max * k_seq;
  const float scale = 1.0f / sqrtf((float)HD);
  int s_qkv, s_o, s_d;
  decode_splits(m, nseq, s_qkv, s_o, s_d);
  CHECK(hwocr_embed_splice(gs->cur_ids, nullptr, m->embed, nullptr, ws->h, nseq, Hd, G ? m->embed_scale : 1.0f, st));
  if (decode_takes_rows16(m, nseq)) {
    // ---- at most 16 reads in flight: 5 launches per layer (csrc/gemm_rows16.hip; 6 with the split attention's merge launch).  The residual stream alternates between
    // ws->h and ws->hn: a layer's QKV projection reads the previous layer's stream + the down projection's slabs in its norm
    // prologue (every workgroup) and writes the updated stream to the OTHER buffer (one workgroup).
    const int sd16 = rows16_down_split(m);
    void* hbuf[2] = {ws->h, ws->hn};
    int cur = 0;
    for (int l = 0; l < m->layers; ++l) {
      const hwocr_dec_layer& L = m->L[l];
      bf16* Kc = B(kv->k) + l * k_layer;
      bf16* Vc = B(kv->vt) + l * k_layer;
      hwocr_rows16_norm n1{hbuf[cur], hbuf[cur ^ 1], Hd, l ? ws->slabs : nullptr, l ? sd16 : 0, (long)nseq * Hd, Hd, L.in_norm_w, m->eps, G};
      cur ^= 1;
      // the QKV projection leaves ONE fp32 slab for the attention launch, BEHIND the down projection's sd16 slabs in ws->slabs (which
      // the next layer's norm prologue still reads): (sd16 * hidden + QW) * nseq floats in all - hwocr_decode_slab_floats
      float* qkv_slab = ws->slabs + (long)sd16 * nseq * Hd;
      CHECK(hwocr_gemm_rows16(nullptr, 0, L.qkv_wt, qkv_slab, QW, nseq, QW, Hd, HWOCR_EPI_PARTIAL, 1, &n1, st));
      CHECK(decode_attention(m, ws, kv, gs, l, qkv_slab, 1, L.qkv_b, decode_lastwg() ? ws->arrive : nullptr, nseq, attn_splits, st));
      CHECK(hwocr_gemm_rows16(ws->attn, OW, L.o_wt, hbuf[cur], Hd, nseq, Hd, OW, HWOCR_EPI_RESIDUAL, 1, nullptr, st));
      hwocr_rows16_norm n2{hbuf[cur], nullptr, Hd, nullptr, 0, 0, 0, L.post_norm_w, m->eps, G};
      CHECK(hwocr_gemm_rows16(nullptr, 0, L.gate_up_wt, ws->act, m->inter, nseq, 2 * m->inter, Hd, G ? HWOCR_EPI_GEGLU : HWOCR_EPI_SWIGLU, 1,
                              &n2, st));
      CHECK(hwocr_gemm_rows16(ws->act, m->inter, L.down_wt, ws->slabs, Hd, nseq, Hd, m->inter, HWOCR_EPI_PARTIAL, sd16, nullptr, st));
    }
    // last layer's slabs + residual -> final norm (the stream's last update is not needed afterwards, but the kernel does it in place)
    CHECK(hwocr_add_rmsnorm(ws->slabs, sd16, (long)nseq * Hd, Hd, nullptr, hbuf[cur], Hd, m->final_norm_w, hbuf[cur ^ 1], Hd, nullptr, nseq,
                            Hd, m->eps, G, st));
    CHECK(decode_gemm(hbuf[cur ^ 1], m->lm_head, m->lm_head_t, m->lm_head8t.w, m->lm_head8t.scale, ws->logits, nseq, m->vocab, Hd, m->vocab,
                      HWOCR_EPI_LINEAR, 1, st));
    return decode_select(m, ws, gs, nseq, st);
  }
  CHECK(hwocr_add_rmsnorm(nullptr, 0, 0, 0, nullptr, ws->h, Hd, m->L[0].in_norm_w, ws->hn, Hd, nullptr, nseq, Hd,
                          m->eps, G, st));
  for (int l = 0; l < m->layers; ++l) {
    const hwocr_dec_layer& L = m->L[l];
    bf16* Kc = B(kv->k) + l * k_layer;
    bf16* Vc = B(kv->vt) + l * k_layer;
    CHECK(decode_gemm(ws->hn, L.qkv_w, L.qkv_wt, L.qkv8t, L.qkv8.scale, ws->slabs, nseq, QW, Hd, QW, HWOCR_EPI_PARTIAL, s_qkv, st));
    // bias + rotary + cache append of this step's q / k / v ride in the attention launch (one launch per layer less;
    // HWOCR_DECODE_FUSE_QKV=0: the two separate launches, for A/B runs)
    static const bool fuse_qkv = HWOCR_DIAG_ENV_INT("HWOCR_DECODE_FUSE_QKV", 1) != 0;
    if (!fuse_qkv) {
      if (kv->fp8) return HWOCR_EINVAL;  // (the E4M3 cache exists in the fused form only)
      CHECK(hwocr_decode_qkv_finish(ws->slabs, s_qkv, (long)nseq * QW, L.qkv_b, ws->q, Kc, Vc, gs->lens, gs->rope_delta,
                                    m->rope_cos, m->rope_sin, nseq, m->Hq, m->Hkv, k_seq, k_head, k_seq, k_head,
                                    kv->ctx, HD, kv->tiled, kv->ctx, m->max_pos, gs->status, st));
      CHECK(hwocr_attn_decode(ws->q, Kc, Vc, gs->lens, ws->attn, ws->part_o, ws->part_ml, decode_lastwg() ? ws->arrive : nullptr, nseq, m->Hq, m->Hkv,
                              attn_splits, k_seq, k_head, k_seq, k_head, kv->ctx, scale, HD, kv->tiled, st));
    } else
    CHECK(decode_attention(m, ws, kv, gs, l, ws->slabs, s_qkv, L.qkv_b, decode_lastwg() ? ws->arrive : nullptr, nseq, attn_splits, st));
    CHECK(decode_gemm(ws->attn, L.o_w, L.o_wt, L.o8t, L.o8.scale, ws->slabs, nseq, Hd, OW, Hd, HWOCR_EPI_PARTIAL, s_o, st));
    CHECK(hwocr_add_rmsnorm(ws->slabs, s_o, (long)nseq * Hd, Hd, nullptr, ws->h, Hd, L.post_norm_w, ws->hn, Hd,
                            nullptr, nseq, Hd, m->eps, G, st));
    CHECK(decode_gemm(ws->hn, L.gate_up_w, L.gate_up_wt, L.gate_up8t, L.gate_up8.scale, ws->act, nseq, 2 * m->inter, Hd, m->inter,
                      G ? HWOCR_EPI_GEGLU : HWOCR_EPI_SWIGLU, 1, st));
    CHECK(decode_gemm(ws->act, L.down_w, L.down_wt, L.down8t, L.down8.scale, ws->slabs, nseq, Hd, m->inter, Hd, HWOCR_EPI_PARTIAL, s_d,
                      st));
    const void* next_norm = (l + 1 < m->layers) ? m->L[l + 1].in_norm_w : m->final_norm_w;
    CHECK(hwocr_add_rmsnorm(ws->slabs, s_d, (long)nseq * Hd, Hd, nullptr, ws->h, Hd, next_norm, ws->hn, Hd, nullptr,
                            nseq, Hd, m->eps, G, st));
  }
  CHECK(decode_gemm(ws->hn, m->lm_head, m->lm_head_t, m->lm_head8t.w, m->lm_head8t.scale, ws->logits, nseq, m->vocab, Hd, m->vocab,
                    HWOCR_EPI_LINEAR, 1, st));
  return decode_select(m, ws, gs, nseq, st);
}

struct DecodeGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int hwocr_decode_graph_create(const hwocr_decoder* m, const hwocr_dec_ws* ws, const hwocr_kv* kv,
                                         const hwocr_gen_state* gs, int nseq, int attn_splits, void** graph_out) {
  if (!graph_out) return HWOCR_EINVAL;
  hipStream_t cap;
  if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) return HWOCR_ELAUNCH;
  DecodeGraph* g = new DecodeGraph();
  int rc = HWOCR_OK;
  if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = HWOCR_ELAUNCH;
  if (rc == HWOCR_OK) {
    rc = hwocr_decode_step(m, ws, kv, gs, nseq, attn_splits, cap);
    if (hipStreamEndCapture(cap, &g->graph) != hipSuccess) rc = rc ? rc : HWOCR_ELAUNCH;
  }
  if (rc == HWOCR_OK && hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) rc = HWOCR_ELAUNCH;
  (void)hipStreamDestroy(cap);
  if (rc != HWOCR_OK) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return rc;
  }
  *graph_out = g;
  return HWOCR_OK;
}

extern "C" int hwocr_decode_graph_launch(void* graph, int n, hipStream_t st) {
  DecodeGraph* g = (DecodeGraph*)graph;
  if (!g || !g->exec || n < 0) return HWOCR_EINVAL;
  for (int i = 0; i < n; ++i)
    if (hipGraphLaunch(g->exec, st) != hipSuccess) return HWOCR_ELAUNCH;
  return HWOCR_OK;
}

extern "C" int hwocr_decode_graph_destroy(void* graph) {
  DecodeGraph* g = (DecodeGraph*)graph;
  if (!g) return HWOCR_EINVAL;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return HWOCR_OK;
}
