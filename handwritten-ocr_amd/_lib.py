"""ctypes binding of include/hwocr.h.

There is no CPU fallback for the device functions: if libhwocr_hip.so is missing or a launch fails this module
raises.  The text library (host C++) is bound separately so that compare/merge work on a machine without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

SELECT_WS_INTS = 40  # HWOCR_SELECT_WS_INTS
ABI_VERSION = 14  # == HWOCR_ABI_VERSION of include/hwocr.h; hip() refuses a library that reports another one

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float


class HwocrError(RuntimeError):
    pass


STATUS_BAD_POSITION = 1  # HWOCR_STATUS_BAD_POSITION

_CODES = {1: "HWOCR_EINVAL (argument rejected by the launcher)", 2: "HWOCR_ELAUNCH (HIP launch failed)"}


def check(rc: int, what: str = "") -> None:
    if os.environ.get("HWOCR_DEBUG_SYNC"):  # diagnosis: surface an asynchronous device fault at the call that caused it
        import sys

        import torch

        sys.stderr.write(f"[hwocr] {what or 'call'} rc={rc}\n")
        sys.stderr.flush()
        torch.cuda.synchronize()
    if rc != 0:
        detail = ""
        if rc == 2 and _hip is not None:
            detail = " — " + (_hip.hwocr_last_error() or b"").decode(errors="replace")
        raise HwocrError(f"{what or 'hwocr call'} failed: {_CODES.get(rc, rc)}{detail}")


class W8(C.Structure):
    """hwocr_w8: E4M3 copy of a weight matrix + one fp32 scale per row (both NULL: the GEMM stays bf16)."""
    _fields_ = [("w", P), ("scale", P)]


class VitBlock(C.Structure):
    _fields_ = [(n, P) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b",
                                  "fc1_w", "fc1_b", "fc2_w", "fc2_b")] + [("windowed", I)] + [
        (n, W8) for n in ("qkv8", "proj8", "fc18", "fc28")]


class Vit(C.Structure):
    _fields_ = [(n, I) for n in ("depth", "dim", "heads", "mlp_dim", "patch", "merge", "tps", "kpad", "out_dim",
                                 "kind", "head_pad")] + [
        ("eps", F), ("patch_w", P), ("blocks", C.POINTER(VitBlock)),
        ("merger_ln_w", P), ("merger_ln_b", P), ("merger_fc1_w", P), ("merger_fc1_b", P),
        ("merger_fc2_w", P), ("merger_fc2_b", P), ("rope_cos", P), ("rope_sin", P), ("pixel_lut", P), ("patch_b", P),
        ("pos_embed", P), ("qk_interleaved", I)]


class VitSplit(C.Structure):
    """hwocr_vit_split: destinations and rotary tables of hwocr_gemm_vit_qkv."""
    _fields_ = [(n, P) for n in ("Q", "K", "VT", "pos_h", "pos_w", "cos_tab", "sin_tab")] + [("heads", I), ("hd", I), ("tok_ld", I)]


class VitLayout(C.Structure):
    _fields_ = [(n, P) for n in ("pos_h", "pos_w", "seg_lens", "row_src", "win_off", "win_lens")] + [
        ("nwin", I), ("max_win", I)]


class VitWs(C.Structure):
    _fields_ = [(n, P) for n in ("patches", "x", "xn", "qkv", "q", "k", "vt", "attn", "mlp", "merge_mid", "q8", "q8s")]


class DecLayer(C.Structure):
    _fields_ = [(n, P) for n in ("in_norm_w", "qkv_w", "qkv_b", "o_w", "post_norm_w", "gate_up_w", "down_w",
                                  "qkv_wt", "o_wt", "gate_up_wt", "down_wt")] + [
        (n, W8) for n in ("qkv8", "o8", "gate_up8", "down8")] + [(n, P) for n in ("qkv8t", "o8t", "gate_up8t", "down8t")]


class Decoder(C.Structure):
    _fields_ = [(n, I) for n in ("layers", "hidden", "Hq", "Hkv", "inter", "vocab", "sec0", "sec1", "head_dim",
                                 "gemma")] + [
        ("eps", F), ("embed_scale", F), ("embed", P), ("lm_head", P), ("lm_head_t", P), ("lm_head8t", W8), ("final_norm_w", P), ("L", C.POINTER(DecLayer)),
        ("rope_cos", P), ("rope_sin", P), ("max_pos", I)]


class Kv(C.Structure):
    _fields_ = [("k", P), ("vt", P), ("nseq_max", I), ("ctx", I), ("tiled", I), ("k_scale", P), ("v_scale", P), ("fp8", I)]


class DecWs(C.Structure):
    _fields_ = [(n, P) for n in ("h", "hn", "qkv", "q", "attn", "act", "slabs", "part_o", "part_ml", "arrive", "select_ws", "logits",
                                  "q8", "q8s", "kt", "vtt")]


class Rows16Norm(C.Structure):
    """hwocr_rows16_norm: the RMSNorm prologue of hwocr_gemm_rows16."""
    _fields_ = [("h_in", P), ("h_out", P), ("ldh", I), ("slabs", P), ("nslab", I), ("slab_stride", L), ("ld_slab", I), ("norm_w", P),
                ("eps", F), ("gemma", I)]


class GenState(C.Structure):
    _fields_ = [(n, P) for n in ("cur_ids", "lens", "n_gen", "finished", "out_tokens", "rope_delta")] + [
        ("max_new", I), ("min_new", I), ("n_eos", I), ("pad_id", I), ("eos", I * 4), ("seen", P), ("seen_ld", I),
        ("rep_penalty", F), ("status", P), ("do_sample", I), ("temperature", F), ("top_k", I), ("top_p", F), ("seed", C.c_ulonglong), ("read_ids", P)]


_HIP_SIGS = {
    "hwocr_abi_version": ([], I),
    "hwocr_last_error": ([], C.c_char_p),
    "hwocr_gemm_wide": ([P, P, P, P, P, I, I, I, I, I, I, I, I, P], I),
    "hwocr_quant_rows_fp8": ([P, P, P, I, I, I, I, P], I),
    "hwocr_layernorm_fp8": ([P, P, P, P, P, I, I, I, I, F, P], I),
    "hwocr_rmsnorm_fp8": ([P, I, P, P, P, I, I, I, F, I, P], I),
    "hwocr_gemm_wide_fp8": ([P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P], I),
    "hwocr_gemm_skinny": ([P, P, P, P, I, I, I, I, I, I, I, I, I, P], I),
    "hwocr_gemm_rows16": ([P, I, P, P, I, I, I, I, I, I, C.POINTER(Rows16Norm), P], I),
    "hwocr_gemm_skinny_variant": ([I, I, I, I, I, I, C.c_char_p, I], I),
    "hwocr_attn_decode_variant": ([I, I, I, C.c_char_p, I], I),
    "hwocr_stream_create_cumask": ([P, I, C.POINTER(C.c_void_p)], I),
    "hwocr_stream_destroy": ([P], I),
    "hwocr_set_cu_budget": ([I], I),
    "hwocr_probe_placement": ([P, I, L, P], I),
    "hwocr_plan_begin": ([], I),
    "hwocr_plan_end": ([C.c_char_p, I, C.POINTER(I)], I),
    "hwocr_tile_weights": ([P, P, I, I, I, P], I),
    "hwocr_tile_weights_fp8": ([P, P, I, I, I, P], I),
    "hwocr_gemm_skinny_w8": ([P, P, P, P, P, I, I, I, I, I, I, I, P], I),
    "hwocr_attn_prefill": ([P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, L, L, L, L, L, L, L, F, I, P], I),
    "hwocr_attn_decode": ([P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, L, F, I, I, P], I),
    "hwocr_attn_decode_qkv": ([P, I, L, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, L, F, I, I, I, I, P, P], I),
    "hwocr_attn_decode_qkv_fp8kv": ([P, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, I, I, P, P], I),
    "hwocr_kv_quant_fp8": ([P, P, L, L, L, L, L, P, P, P, P, I, I, I, I, P], I),
    "hwocr_attn_varlen": ([P, P, P, P, P, P, I, I, I, I, L, L, L, L, L, L, L, F, P], I),
    "hwocr_patchify": ([P, P, P, I, I, I, I, I, I, I, I, P, P], I),
    "hwocr_layernorm": ([P, P, P, P, I, I, I, I, F, P], I),
    "hwocr_add_rmsnorm": ([P, I, L, I, P, P, I, P, P, I, P, I, I, F, I, P], I),
    "hwocr_vit_rope_split": ([P, P, P, P, P, P, P, P, I, I, I, I, I, P], I),
    "hwocr_gemm_vit_qkv": ([P, P, P, I, I, I, I, P, P, C.POINTER(VitSplit), P], I),
    "hwocr_mrope_kv_prefill": ([P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, L, I, I, P], I),
    "hwocr_decode_qkv_finish": ([P, I, L, P, P, P, P, P, P, P, P, I, I, I, L, L, L, L, L, I, I, I, I, P, P], I),
    "hwocr_embed_splice": ([P, P, P, P, P, I, I, F, P], I),
    "hwocr_argmax_advance": ([P, I, I, I, P, P, P, P, P, I, I, C.POINTER(I), I, I, P, I, F, P, P], I),
    "hwocr_sample_advance": ([P, I, I, I, P, P, P, P, P, I, I, C.POINTER(I), I, I, P, I, F, F, I, F, C.c_ulonglong, P, P, P], I),
    "hwocr_vit_forward": ([C.POINTER(Vit), C.POINTER(VitWs), P, I, I, I, I, C.POINTER(VitLayout), P, P], I),
    "hwocr_prefill": ([C.POINTER(Decoder), C.POINTER(DecWs), C.POINTER(Kv), C.POINTER(GenState), P, P, P, P, P, P,
                       I, I, I, I, P], I),
    "hwocr_decode_step": ([C.POINTER(Decoder), C.POINTER(DecWs), C.POINTER(Kv), C.POINTER(GenState), I, I, P], I),
    "hwocr_decode_slab_floats": ([C.POINTER(Decoder), I], C.c_long),
    "hwocr_decode_graph_create": ([C.POINTER(Decoder), C.POINTER(DecWs), C.POINTER(Kv), C.POINTER(GenState), I, I,
                                   C.POINTER(P)], I),
    "hwocr_decode_graph_launch": ([P, I, P], I),
    "hwocr_decode_graph_destroy": ([P], I),
    "hwocr_img_luma_sum": ([P, L, P, P], I),
    "hwocr_img_contrast": ([P, P, L, I, F, P], I),
    "hwocr_img_contrast_dev": ([P, P, L, P, L, F, P], I),
    "hwocr_img_binarize": ([P, P, L, P], I),
    "hwocr_img_sharpen": ([P, P, I, I, P], I),
    "hwocr_img_resize_bicubic": ([P, P, P, I, I, I, I, P, P, I, P, P, I, P], I),
    "hwocr_profile_enable": ([I], I),
    "hwocr_profile_read": ([C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)], I),
}

_TEXT_SIGS = {
    "hwocr_levenshtein_u32": ([P, C.c_int64, P, C.c_int64], C.c_int64),
    "hwocr_lcs_align_u32": ([P, C.c_int64, P, C.c_int64, P], I),
}

HIP_SYMBOLS = tuple(_HIP_SIGS)
TEXT_SYMBOLS = tuple(_TEXT_SIGS)

_hip = None
_text = None


def _bind(path: str, sigs: dict) -> C.CDLL:
    lib = C.CDLL(path)
    for name, (argtypes, restype) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def hip() -> C.CDLL:
    """The device library.  Raises if it has not been built (no fallback path exists)."""
    global _hip
    if _hip is None:
        # torch bundles its own libamdhip64; it must be in the process BEFORE our library is loaded, otherwise the
        # loader resolves our NEEDED libamdhip64.so.7 to the system copy and the process ends up with two HIP runtimes
        # (streams and pointers of one are "no device" to the other).
        import torch  # noqa: F401

        if os.environ.get("HWOCR_DIAG_LIB", "0") not in ("", "0"):
            # measurement runs only (tools/ab_env.sh): the -DHWOCR_DIAG build, in which the A/B switches of concluded experiments exist
            import sys

            sys.stderr.write("hwocr: HWOCR_DIAG_LIB set - loading the DIAGNOSTIC library (A/B switches live; not the shipped code)\n")
            _build.use_diag_library()
        if not os.path.exists(_build.HIP_LIB):
            raise HwocrError(
                f"{_build.HIP_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "the page-read engine has no CPU fallback")
        lib = _bind(_build.HIP_LIB, _HIP_SIGS)
        if lib.hwocr_abi_version() != ABI_VERSION:  # a stale build: struct layouts would not match
            raise HwocrError(f"{_build.HIP_LIB} reports ABI {lib.hwocr_abi_version()}, these bindings are for {ABI_VERSION}: "
                             "rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        _hip = lib
    return _hip


def text() -> C.CDLL:
    global _text
    if _text is None:
        if not os.path.exists(_build.TEXT_LIB):
            _build.build_text()
        _text = _bind(_build.TEXT_LIB, _TEXT_SIGS)
    return _text


def ptr(t) -> C.c_void_p:
    """Device/host pointer of a torch tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_handle():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
