"""Every kernel instance — and every size-dependent code path inside one — that the WIDE half of a page read (vision tower +
prefill: 63 % of the bench's step) dispatches must have an oracle case (the persistent multi-tile loop of
gemm_wide256_kernel and the grid-stride loop of layernorm_kernel ran only inside bench.py).

The launch list is not restated here: hwocr_vit_forward / hwocr_prefill are run under the library's plan recording
(hwocr_plan_begin: every launcher validates its arguments, notes "<kernel instance> <geometry>" and returns without touching the
device — engine.wide_plan), at the launch geometry of `python bench.py` and of the other legs (single page, the test engines),
for every preset.  Each line is reduced to a CLASS — the instance plus what changes the code path inside it (more tiles than
workgroups, more rows than one grid trip, slabs / gather / fp8 output ...) — and must be among the classes of the parity cases of
tests/test_ops_gpu.py, computed the same way from their shapes.  No GPU needed."""
import ctypes as C
import re

import pytest

from handwritten_ocr_amd import _lib, engine
from tests import test_decode_variants as dv
from tests import test_ops_gpu as ops

ONE = C.c_void_p(64)


def _plan(call):
    """Lines the launchers note for `call(lib)` under plan recording."""
    lib = _lib.hip()
    assert lib.hwocr_plan_begin() == 0
    try:
        rc = call(lib)
    finally:
        need = C.c_int()
        lib.hwocr_plan_end(None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    assert rc == 0, rc
    return [l for l in buf.value.decode().split("\n") if l]


def _fields(line):
    return {k: v for k, v in re.findall(r"(\w+)=([^\s,<>]+)", line.split(" ", 1)[1] if " " in line else "")}


def klass(line: str) -> tuple:
    """Kernel instance + the geometry bits that select a code path inside it."""
    inst = line.split(" ", 1)[0] if not line.startswith(("gemm_skinny_kernel", "gemm_stream")) else line.split(" rows=")[0]
    f = _fields(line)
    if inst.startswith("gemm_wide256_kernel"):
        return (inst, "several tiles per workgroup" if int(f["rounds"]) > 1 else "one tile per workgroup")
    if inst.startswith("gemm_wide256w4_kernel"):  # + the K loop's forms: 2 K tiles, 3 (no steady-state trip), more
        return (inst, "several tiles per workgroup" if int(f["rounds"]) > 1 else "one tile per workgroup", "ktiles " + (f["ktiles"] if int(f["ktiles"]) <= 3 else ">3"))
    if inst.startswith("layernorm_kernel"):
        return (inst, "fp8=" + f["fp8"], "several grid trips" if int(f["trips"]) > 1 else "one trip")
    if inst.startswith("add_rmsnorm_kernel"):
        return (inst, "fp8=" + f["fp8"], "gemma=" + f["gemma"], "slabs" if int(f["nslab"]) else "no slabs", "gather=" + f["gather"])
    if inst.startswith("add_rmsnorm_row_kernel"):
        return (inst, "gemma=" + f["gemma"], "slabs" if int(f["nslab"]) else "no slabs", "gather=" + f["gather"])
    if inst.startswith("attn_prefill_kernel"):
        return (inst, "tiled=" + f["tiled"], "varlen=" + f["varlen"])
    if inst.startswith("patchify_kernel"):
        return (inst, "permuted=" + f["permuted"])
    if inst.startswith("vit_rope_split_kernel"):
        return (inst, "interleaved=" + f["interleaved"])
    if inst.startswith("mrope_kv_prefill_kernel"):
        return (inst, "tiled=" + f["tiled"])
    if inst.startswith("embed_splice_kernel"):
        return (inst, "spliced=" + f["spliced"])
    if inst.startswith("argmax_advance_kernel"):
        return (inst,)
    return (inst,)


# ------------------------------------------------------------------------------------------------ classes of the parity cases
def _gemm(M, N, K, epi):
    ldo = N // 2 if epi in (4, 7) else N
    return _plan(lambda lib: lib.hwocr_gemm_wide(ONE, ONE, ONE, ONE if epi == 1 else None, ONE, M, N, K, K, K, ldo, N if epi == 1 else 0,
                                                 epi, None))


def _gemm8(M, N, K, epi):
    ldo = N // 2 if epi in (4, 7) else N
    return _plan(lambda lib: lib.hwocr_gemm_wide_fp8(ONE, ONE, ONE, ONE, ONE, ONE if epi == 1 else None, ONE, M, N, K, K, K, ldo,
                                                     N if epi == 1 else 0, epi, None))


def _vit_qkv(M, K, heads, hd, fp8):
    sp = _lib.VitSplit(Q=ONE, K=ONE, VT=ONE, pos_h=ONE, pos_w=ONE, cos_tab=ONE, sin_tab=ONE, heads=heads, hd=hd, tok_ld=(M + 63) // 64 * 64)
    return _plan(lambda lib: lib.hwocr_gemm_vit_qkv(ONE, ONE, ONE, M, K, K, K, ONE if fp8 else None, ONE if fp8 else None, C.byref(sp), None))


def _attn(hd, group, causal, tiled, max_len, nseg=4, heads=8):
    return _plan(lambda lib: lib.hwocr_attn_prefill(ONE, ONE, ONE, ONE, ONE, nseg, heads, group, hd, max_len, causal, 8, 8, 8, 8, 8, 8, 8, 8,
                                                    8, 8, 8, 1.0, tiled, None))


def covered() -> set:
    lines = []
    for M, N, K in ops.WIDE_GEMM_SHAPES:
        for epi in ops.WIDE_GEMM_EPIS:
            lines += _gemm(M, N, K, epi)
    for M, N, K, epi in ops.WIDE_BENCH_CASES + ops.W4_LOOP_CASES:
        lines += _gemm(M, N, K, epi)
    for M, N, K in ops.WIDE_SWIGLU_SHAPES:
        lines += _gemm(M, N // 32 * 32, K, 4)
    for M, N, K in ops.WIDE_GEGLU_SHAPES:
        lines += _gemm(M, N, K, 7)
    for M, N, K in ops.WIDE_FP8_SHAPES:
        for epi in ops.WIDE_FP8_EPIS:
            lines += _gemm8(M, N, K, epi)
    for M, N, K, epi in ops.WIDE_FP8_BENCH_CASES:
        lines += _gemm8(M, N, K, epi)
    for M, N, K in ops.WIDE_FP8_GATED_SHAPES:
        lines += _gemm8(M, N, K, 4) + _gemm8(M, N, K, 7)
    for M, K, heads, hd in ops.VIT_QKV_CASES:
        lines += _vit_qkv(M, K, heads, hd, 0)
        lines += _gemm(M, 3 * heads * hd, K, 0)   # the two-launch form the fused one is held to
        lines += _plan(lambda lib: lib.hwocr_vit_rope_split(ONE, ONE, ONE, ONE, ONE, ONE, ONE, ONE, M, (M + 63) // 64 * 64, heads, hd, 1, None))
        if K % 128 == 0:
            lines += _vit_qkv(M, K, heads, hd, 1)
    for rows, D in ops.LAYERNORM_CASES:
        lines += _plan(lambda lib: lib.hwocr_layernorm(ONE, ONE, ONE, ONE, rows, D, D, D, 1e-6, None))
    for rows, D in ops.NORM_FP8_CASES:
        lines += _plan(lambda lib: lib.hwocr_layernorm_fp8(ONE, ONE, ONE, ONE, ONE, rows, D, D, D, 1e-6, None))
        lines += _plan(lambda lib: lib.hwocr_layernorm(ONE, ONE, ONE, ONE, rows, D, D, D, 1e-6, None))
        for g in (0, 1):
            lines += _plan(lambda lib: lib.hwocr_rmsnorm_fp8(ONE, D, ONE, ONE, ONE, D, rows, D, 1e-6, g, None))
            lines += _plan(lambda lib: lib.hwocr_add_rmsnorm(None, 0, 0, 0, None, ONE, D, ONE, ONE, D, None, rows, D, 1e-6, g, None))
        lines += _plan(lambda lib: lib.hwocr_quant_rows_fp8(ONE, ONE, ONE, rows, D, D, D, None))
    for rows, D, nslab in ops.ADD_RMSNORM_CASES:
        for g in (0, 1):
            lines += _plan(lambda lib: lib.hwocr_add_rmsnorm(ONE if nslab else None, nslab, rows * D, D, ONE if nslab else None, ONE, D, ONE,
                                                             ONE, D, None, rows, D, 1e-6, g, None))
    for D, g in ops.ADD_RMSNORM_GATHER_CASES:
        lines += _plan(lambda lib: lib.hwocr_add_rmsnorm(None, 0, 0, 0, None, ONE, D, ONE, ONE, D, ONE, 3, D, 1e-6, g, None))
    for hd, Hq, Hkv, causal, tiled in ops.ATTN_PREFILL_CASES:
        lines += _attn(hd, Hq // Hkv, int(causal), tiled, 300, heads=Hq)
    for lens in ops.ATTN_VIT80_LONG_LENS:                 # test_attn_vit80_long_segments walks the x / 12 / 4 forms by env var;
        lines += _attn(80, 1, 0, 0, max(lens))            # here: the launcher's own choice for those lengths
    lines += _attn(80, 1, 0, 0, 5184)                     # test_attn_vit80_page_shape
    lines += _attn(256, 8, 0, 0, max(ops.ATTN_HD256_LENS), heads=8)   # test_attn_prefill_hd256_long_reads
    for hd, heads in ops.ATTN_VARLEN_CASES:
        lines += _plan(lambda lib: lib.hwocr_attn_varlen(ONE, ONE, ONE, ONE, ONE, ONE, 10, heads, hd, 64, 8, 8, 8, 8, 8, 8, 8, 1.0, None))
    for permuted in (0, 1):                               # test_patchify_exact
        lines += _plan(lambda lib: lib.hwocr_patchify(ONE, ONE, ONE, 1, 56, 84, 14, 2, 2, 1216, 64, ONE if permuted else None, None))
    for hd, _s0, _s1, tiled in ops.MROPE_CASES:
        lines += _plan(lambda lib: lib.hwocr_mrope_kv_prefill(ONE, ONE, ONE, ONE, ONE, ONE, ONE, 128, 64, 4, 2, 16, 40, 8, 8, 8, 8, 64, hd, tiled,
                                                              None))
    lines += _plan(lambda lib: lib.hwocr_embed_splice(ONE, ONE, ONE, ONE, ONE, 64, 256, 1.0, None))           # test_embed_splice
    # tests/test_kv_fp8_gpu.py::test_kv_quant_fp8_codes_scales_and_layout (the prefill's fill of the E4M3 KV cache)
    lines += _plan(lambda lib: lib.hwocr_kv_quant_fp8(ONE, ONE, 8, 8, 8, 8, 4160, ONE, ONE, ONE, ONE, 2, 2, 4160, 4224, None))
    eos = (C.c_int * 4)(1, 0, 0, 0)
    lines += _plan(lambda lib: lib.hwocr_argmax_advance(ONE, 512, 512, 4, ONE, ONE, ONE, ONE, ONE, 8, 0, eos, 1, 0, None, 0, 1.0, None, None))
    # test_argmax_* at V = 151936 with split_ws: 16 workgroups per read, the last one finishing
    lines += _plan(lambda lib: lib.hwocr_argmax_advance(ONE, 151936, 151936, 4, ONE, ONE, ONE, ONE, ONE, 8, 0, eos, 1, 0, None, 0, 1.0, ONE, None))
    out = {klass(l) for l in lines}
    # the LM head of a prefill chunk is a decode GEMM at <= 16 rows: those instances are the business of tests/test_decode_variants.py
    gemm, _ = dv._covered()
    out |= {(g,) for g in gemm}
    return out


# ------------------------------------------------------------------------------------------------ what the configurations launch
def _hw(cfg, side):
    from handwritten_ocr_amd import imageproc

    if cfg.family == "paligemma":
        return cfg.image_size, cfg.image_size
    return imageproc.smart_resize(side, side, cfg.patch_size * cfg.merge, cfg.min_pixels, cfg.max_pixels)


# (preset, page side, fp8 engine?)
CONFIGS = [("qwen2-vl-2b", 1024, False), ("qwen2-vl-2b", 1024, True), ("qwen2.5-vl-7b", 1024, False), ("qwen2.5-vl-3b", 1024, False),
           ("paligemma-3b", 896, False), ("paligemma-3b", 896, True), ("small", 512, False), ("tiny", 112, False), ("tiny25", 112, False),
           ("tinypg", 56, False), ("tinypg", 56, True)]
# (pages per tower launch, prompts per prefill launch): python bench.py; its single-page leg; the engines of tests/test_fullsize_gpu.py
GEOMETRIES = [(12, 16), (3, 3), (3, 4), (1, 1)]


@pytest.mark.parametrize("preset,side,fp8", CONFIGS)
def test_every_launch_of_the_wide_half_has_an_oracle_case(preset, side, fp8):
    have = covered()
    cfg = engine.preset(preset)
    H, W = _hw(cfg, side)
    n_img = (H // cfg.patch_size) * (W // cfg.patch_size) // cfg.merge ** 2
    prompt = n_img + (17 if cfg.family == "paligemma" else 32)
    missing = {}
    for pages, reads in GEOMETRIES:
        for line in engine.wide_plan(cfg, (H, W), pages, reads, prompt, fp8):
            k = klass(line)
            if k not in have:
                missing.setdefault(k, f"{pages} pages / {reads} prompts: {line}")
    assert not missing, f"{preset} (fp8={fp8}): launches without a parity case:\n" + "\n".join(f"  {k}: {v}" for k, v in missing.items())


def test_bench_default_wide_gemms_are_literal_cases():
    """`python bench.py` = Qwen2-VL-2B, 12 pages per tower launch, 16 prompts of 1328 tokens per prefill launch: every wide GEMM of
    that step is among the parity cases shape for shape (M, N, K, epilogue), not just class for class."""
    cfg = engine.preset("qwen2-vl-2b")
    literal = {(m, n, k, e) for (m, n, k, e) in ops.WIDE_BENCH_CASES} | {(m, n // 32 * 32, k, 4) for (m, n, k) in ops.WIDE_SWIGLU_SHAPES} | \
              {(m, 3 * h * d, k, 8) for (m, k, h, d) in ops.VIT_QKV_CASES}
    rows = {(r, d) for (r, d) in ops.LAYERNORM_CASES}
    for line in engine.wide_plan(cfg, (1008, 1008), 12, 16, 1328, False):
        f = _fields(line)
        if line.startswith("gemm_wide256_kernel"):
            epi = int(re.search(r"epi=(\d+)", line).group(1))
            assert (int(f["M"]), int(f["N"]), int(f["K"]), epi) in literal, line
        if line.startswith("layernorm_kernel"):
            assert (int(f["rows"]), int(f["D"])) in rows, line


def test_the_plan_sees_the_multi_tile_and_multi_trip_paths_of_the_bench():
    """The premise of this file: at the bench's geometry the persistent GEMM walks several tiles per workgroup and the LayerNorm
    takes several grid trips (if a later change makes that untrue the classes above lose their meaning)."""
    plan = engine.wide_plan(engine.preset("qwen2-vl-2b"), (1008, 1008), 12, 16, 1328, False)
    g = [l for l in plan if l.startswith("gemm_wide256_kernel")]
    assert g and all(int(_fields(l)["rounds"]) > 1 for l in g)
    ln = [l for l in plan if l.startswith("layernorm_kernel")]
    assert ln and all(int(_fields(l)["trips"]) == 4 for l in ln)


def test_plan_recording_launches_nothing_and_rejects_what_the_launchers_reject():
    lib = _lib.hip()
    assert _plan(lambda l: l.hwocr_gemm_wide(ONE, ONE, None, None, ONE, 2048, 512, 128, 128, 128, 512, 0, 0, None)) == \
        ["gemm_wide256w4_kernel<epi=0> M=2048 N=512 K=128 tiles=16 grid=16 rounds=1 ktiles=2"]   # (plain epilogue: the four-wave form)
    assert _plan(lambda l: l.hwocr_gemm_wide(ONE, ONE, ONE, ONE, ONE, 2048, 512, 4096, 4096, 4096, 512, 512, 1, None)) == \
        ["gemm_wide256_kernel<epi=1,stagger,bf16> M=2048 N=512 K=4096 tiles=16 grid=16 rounds=1 ktiles=64"]  # (residual, long K loop, few column tiles)
    assert lib.hwocr_plan_begin() == 0
    assert lib.hwocr_gemm_wide(ONE, ONE, None, None, ONE, 64, 64, 72, 72, 72, 64, 0, 0, None) == 1   # K % 64: still refused
    need = C.c_int()
    assert lib.hwocr_plan_end(None, 0, C.byref(need)) == 1 and need.value == 1                        # nothing was noted
    buf = C.create_string_buffer(4)
    assert lib.hwocr_plan_end(buf, 4, C.byref(need)) == 0 and buf.value == b""


def test_cu_budget_sizes_the_persistent_grid():
    """hwocr_set_cu_budget (a thread's launches go into a CU-masked stream, tools/bench_partition.py): the persistent GEMM's grid
    follows it, and only it; the arguments of the stream calls are checked before anything touches a device."""
    lib = _lib.hip()
    grid = lambda: int(_fields(_plan(lambda l: l.hwocr_gemm_wide(ONE, ONE, None, None, ONE, 62208, 1280, 1280, 1280, 1280, 1280, 0, 0,
                                                                  None))[0])["grid"])  # noqa: E731
    whole = grid()
    assert whole >= 192
    assert lib.hwocr_set_cu_budget(192) == 0
    try:
        assert grid() == 192
        assert lib.hwocr_set_cu_budget(100000) == 0
        assert grid() == whole
    finally:
        assert lib.hwocr_set_cu_budget(0) == 0
    assert grid() == whole
    assert lib.hwocr_set_cu_budget(-1) == 1
    out = C.c_void_p()
    assert lib.hwocr_stream_create_cumask(None, 8, C.byref(out)) == 1
    assert lib.hwocr_stream_create_cumask((C.c_uint * 8)(), 0, C.byref(out)) == 1
    assert lib.hwocr_stream_destroy(None) == 1
    assert lib.hwocr_probe_placement(None, 4, 0, None) == 1


def test_the_four_wave_gemm_forms_all_have_parity_cases():
    """Every code path of gemm_wide256w4_kernel, for every epilogue it is taken for by default, is reached by a parity case: K loop of
    2 / 3 / more K tiles x one / several tiles per workgroup (classes as `klass` draws them)."""
    have = covered()
    w4 = {k for k in have if k[0].startswith("gemm_wide256w4_kernel")}
    for epi in (0, 1, 2, 4):
        inst = f"gemm_wide256w4_kernel<epi={epi}>"
        assert any(k[0] == inst for k in w4), inst
    for kt in ("ktiles 2", "ktiles 3", "ktiles >3"):
        for rounds in ("one tile per workgroup", "several tiles per workgroup"):
            assert any(k[1] == rounds and k[2] == kt for k in w4), (kt, rounds)
