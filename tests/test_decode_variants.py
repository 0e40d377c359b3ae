"""Every kernel instance a shipped configuration's decode step dispatches must have an oracle case.

The decode GEMM launcher picks a kernel instance from the shape (csrc/gemm_stream.hip plan_stream; csrc/gemm.hip
skinny_takes_stream) and so does the decode attention.  This test asks the library's own planner — no launch, no GPU —
which instance every preset runs at the read counts the bench uses, and checks it against the instances reached by the
parity cases of tests/test_ops_gpu.py (which run on the MI355X against the fp32 product of the same operands)."""
import ctypes as C

import pytest

from handwritten_ocr_amd import _lib, engine
from tests import test_ops_gpu as ops

PRESETS = ("qwen2-vl-2b", "qwen2.5-vl-7b", "qwen2.5-vl-3b", "paligemma-3b", "small", "tiny", "tiny25", "tinypg")
READS = (1, 3, 16, 17, 126, 252)


def _gemm_variant(B, N, K, epi, splitk, w_tiled=1):
    buf = C.create_string_buffer(128)
    _lib.check(_lib.hip().hwocr_gemm_skinny_variant(B, N, K, epi, splitk, w_tiled, buf, len(buf)))
    return buf.value.decode()


def _attn_variant(nsplit, hd, tiled):
    buf = C.create_string_buffer(128)
    _lib.check(_lib.hip().hwocr_attn_decode_variant(nsplit, hd, tiled, buf, len(buf)))
    return buf.value.decode()


def _attn_plan_variant(nsplit, hd, tiled, arrive):
    """The instance name hwocr_attn_decode itself notes for a call (plan recording: nothing is launched) - with arrival counters
    (the last workgroup merges: "+lastwg") or without (a merge launch: "+merge").  What the launcher says, not a rewritten string."""
    lib = _lib.hip()
    one = C.c_void_p(64)
    Hq, Hkv, ctx, B = (8, 1, 1280, 4) if hd == 256 else (12, 2, 2048, 4)
    assert lib.hwocr_plan_begin() == 0
    rc = lib.hwocr_attn_decode(one, one, one, one, one, one, one, one if arrive else None, B, Hq, Hkv, nsplit, Hkv * ctx * hd, ctx * hd,
                               Hkv * hd * ctx, hd * ctx, ctx, hd ** -0.5, hd, tiled, None)
    need = C.c_int()
    lib.hwocr_plan_end(None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value)
    lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    assert rc == 0
    return buf.value.decode().split("\n")[0].split(" ")[0]


def _attn_plan_variant_e4m3(nsplit, arrive):
    """... of hwocr_attn_decode_qkv_fp8kv (the fused step over the E4M3 KV cache of 256-wide heads)."""
    lib = _lib.hip()
    one = C.c_void_p(64)
    assert lib.hwocr_plan_begin() == 0
    rc = lib.hwocr_attn_decode_qkv_fp8kv(one, 1, 8, None, one, one, one, one, one, one, one, one, one, one, one, one if arrive else None, 4, 8,
                                         1, nsplit, 1.0, 512, 1024, one, None)
    need = C.c_int()
    lib.hwocr_plan_end(None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value)
    lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    assert rc == 0
    return buf.value.decode().split("\n")[0].split(" ")[0]


def _rows16_variant(B, N, K, epi, splitk, norm, nslab, gemma):
    """The instance + path class hwocr_gemm_rows16 notes for a call (plan recording: nothing is launched)."""
    lib = _lib.hip()
    one = C.c_void_p(64)
    blk = _lib.Rows16Norm(h_in=one, h_out=one, ldh=K, slabs=one if nslab else None, nslab=nslab, slab_stride=B * K, ld_slab=K, norm_w=one,
                          eps=1e-6, gemma=gemma)
    assert lib.hwocr_plan_begin() == 0
    rc = lib.hwocr_gemm_rows16(None if norm else one, 0 if norm else K, one, one, N // 2 if epi in (4, 7) else N, B, N, K, epi, splitk,
                               C.byref(blk) if norm else None, None)
    need = C.c_int()
    lib.hwocr_plan_end(None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value)
    lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    assert rc == 0
    return buf.value.decode().split(" rows=")[0]


def _covered():
    gemm = {_gemm_variant(*c) for c in ops.DECODE_GEMM_CASES}
    gemm |= {_rows16_variant(*c) for c in ops.ROWS16_CASES}
    # the older operator cases: (B, N, K) x {linear, partial} and the gated ones at N = 3584
    for B in (1, 7, 16, 48, 96, 128, 190, 256):
        for N, K in ((96, 64), (2048, 1536), (1536, 2304)):
            gemm.add(_gemm_variant(B, N, K, 0, 1))
            gemm.add(_gemm_variant(B, N, K, 5, 1))
    for B in (3, 48, 96, 252):
        gemm |= {_gemm_variant(B, 3584, 1536, 4, 1), _gemm_variant(B, 3584, 1536, 7, 1)}
    # every attention parity case runs the call without arrival counters (a merge launch behind a split one) and - split cases, in
    # test_ops_gpu._same_with_the_last_workgroup_merging - again WITH them (the last workgroup merges: the form hwocr_decode_step
    # uses); both names come from the launcher's own plan note of exactly those two calls
    shapes = [(ns, hd, tiled) for (_, _, _, hd, tiled, _, ns) in ops.ATTN_DECODE_BENCH_CASES]
    shapes += [(ns, hd, tiled) for ns in (1, 4) for hd, tiled in ((128, 0), (128, 1), (256, 0))]
    attn = set()
    for ns, hd, tiled in shapes:
        attn.add(_attn_plan_variant(ns, hd, tiled, arrive=False))
        assert _attn_variant(ns, hd, tiled) in attn, "hwocr_attn_decode_variant must name what the launcher notes"
        if ns > 1:
            attn.add(_attn_plan_variant(ns, hd, tiled, arrive=True))
    return gemm, attn


@pytest.mark.parametrize("preset", PRESETS)
def test_every_dispatched_decode_kernel_has_an_oracle_case(preset):
    gemm, attn = _covered()
    cfg = engine.preset(preset)
    missing = []
    for reads in READS:
        plan = engine.decode_plan(cfg, reads)
        for name in engine.DECODE_GEMMS:
            if plan[name][4] not in gemm:
                missing.append((reads, name) + plan[name])
        if plan["attn"] not in attn:
            missing.append((reads, "attn", plan["attn"]))
    assert not missing, f"{preset}: decode kernels without a parity case: {missing}"


@pytest.mark.parametrize("preset", ("paligemma-3b", "qwen2-vl-2b", "qwen2.5-vl-7b", "tinypg", "tiny"))
def test_every_dispatched_e4m3_decode_kernel_has_an_oracle_case(preset):
    """The same for engines built with fp8=True (E4M3 decode weights, hwocr_gemm_skinny_w8)."""
    covered = {_gemm_variant(*c, w_tiled=2) for c in ops.DECODE_GEMM_CASES_W8}
    _, attn = _covered()
    # the E4M3 KV cache of 256-wide heads: tests/test_kv_fp8_gpu.py runs the fused step with and without splits / arrival counters
    attn |= {_attn_plan_variant_e4m3(ns, arrive) for ns, arrive in ((1, False), (4, False), (4, True), (3, False), (3, True))}
    cfg = engine.preset(preset)
    missing = []
    for reads in READS:
        plan = engine.decode_plan(cfg, reads, fp8=True)
        if plan["attn"] not in attn:
            missing.append((reads, "attn", plan["attn"]))
        for name in engine.DECODE_GEMMS:
            if plan[name][4] not in covered:
                missing.append((reads, name) + plan[name])
    assert not missing, f"{preset} (fp8): decode kernels without a parity case: {missing}"


def test_bench_default_shapes_are_literal_cases():
    """`python bench.py` = Qwen2-VL-2B at 252 reads: its five GEMMs must be among the cases shape for shape."""
    plan = engine.decode_plan(engine.preset("qwen2-vl-2b"), 252)
    for name in engine.DECODE_GEMMS:
        N, K, epi, splitk, _ = plan[name]
        assert (N, K, epi, splitk) in ops.DECODE_GEMM_SHAPES[252], (name, plan[name])


def test_variant_query_rejects_what_the_launcher_rejects():
    buf = C.create_string_buffer(64)
    lib = _lib.hip()
    assert lib.hwocr_gemm_skinny_variant(257, 2048, 1536, 0, 1, 1, buf, len(buf)) == 1   # > 256 rows
    assert lib.hwocr_gemm_skinny_variant(8, 2048, 1536, 0, 2, 1, buf, len(buf)) == 1     # split-K without PARTIAL
    assert lib.hwocr_attn_decode_variant(17, 128, 1, buf, len(buf)) == 1
    assert lib.hwocr_attn_decode_variant(1, 256, 1, buf, len(buf)) == 1                  # tiled cache is head_dim 128 only


@pytest.mark.parametrize("preset", PRESETS)
def test_the_slab_buffer_contract(preset):
    """hwocr_dec_ws.slabs (hwocr.h): a decode step writes hwocr_decode_slab_floats(m, nseq) fp32 there.  Held here (host only) to
    what the step's own plan says - split-K slabs of the widest slab-producing GEMM; at <= 16 reads the down projection's slabs
    with the QKV slab behind them (that slab used to go to ws->qkv, a bf16 buffer sized for bf16) - and to the
    engine's allocation rule (engine.ReadEngine._dec_ws: 40 * max_reads * max(QW, hidden))."""
    cfg = engine.preset(preset)
    QW = (cfg.q_heads + 2 * cfg.kv_heads) * cfg.head_dim
    for fp8 in (False, True):
        for reads in READS + (2, 24, 128, 129, 256):
            plan = engine.decode_plan(cfg, reads, fp8=fp8)
            need = engine.decode_slab_floats(cfg, reads, fp8=fp8)
            rows16 = "gemm_rows16_kernel" in plan["qkv"][4]
            if rows16:
                assert need == (plan["down"][3] * cfg.hidden + QW) * reads, (reads, plan["down"], need)
            else:
                want = max(plan["qkv"][3] * QW, plan["o"][3] * cfg.hidden, plan["down"][3] * cfg.hidden) * reads
                assert need == want, (reads, plan["qkv"], plan["o"], plan["down"], need)
            for max_reads in {reads, 252}:
                assert need <= 40 * max_reads * max(QW, cfg.hidden)
    assert _lib.hip().hwocr_decode_slab_floats(None, 4) == -1
