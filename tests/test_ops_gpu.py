"""Operator-level parity on the MI355X: each HIP kernel, called through the C ABI, against the fp32 restatement of
the op it replaces (same bf16 inputs, fp32 math, the reference's rounding points).  Tolerances are in bf16 ulps of
the output and are written at each assert."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._gpu_util import (DEV, assert_close_bf16, assert_close_bf16_explained, bf16_neighbours, lib, p, randbf, randf32,  # noqa: E402
                             rbf, st, tile_k, tile_v, untile_k, untile_v)


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ GEMM
def _epilogue_ref(acc, bias, res, epi):
    v = acc + (bias.float() if bias is not None else 0.0)
    if epi == 0:
        return v
    if epi == 1:
        return rbf(v) + res.float()
    if epi == 2:
        x = rbf(v)
        t = rbf(1.702 * x)
        return x * rbf(torch.sigmoid(t))
    if epi == 3:
        x = rbf(v)
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if epi == 6:
        return torch.nn.functional.gelu(rbf(v), approximate="tanh")
    raise AssertionError


def _activation_candidates(v, epi):
    """Outcomes of an activation epilogue (epi 2 / 3 / 6) at fp32 pre-activations `v` [n] with every rounded intermediate moved by
    -1 / 0 / +1 bf16 ulp: x = bf16(acc + bias), and for quick-GELU the rounded gate bf16(sigmoid(bf16(1.702 x))).  [n, 3 or 9]."""
    xs = bf16_neighbours(rbf(v))                                           # [n, 3]
    if epi == 2:
        s = bf16_neighbours(rbf(torch.sigmoid(rbf(1.702 * xs))))           # [n, 3, 3]
        return (xs.unsqueeze(-1) * s).flatten(1)
    if epi == 3:
        return 0.5 * xs * (1.0 + torch.erf(xs / math.sqrt(2.0)))
    return torch.nn.functional.gelu(xs, approximate="tanh")


def _gated_candidates(g_pre, u_pre, geglu):
    """Outcomes of a gated epilogue at fp32 gate / up pre-activations [n]: gate, activation and up each moved by -1 / 0 / +1 ulp."""
    gs = bf16_neighbours(rbf(g_pre))                                       # [n, 3]
    act = torch.nn.functional.gelu(gs, approximate="tanh") if geglu else torch.nn.functional.silu(gs)
    acts = bf16_neighbours(rbf(act))                                       # [n, 3, 3]
    us = bf16_neighbours(rbf(u_pre))                                       # [n, 3]
    return (acts.flatten(1).unsqueeze(-1) * us.unsqueeze(1)).flatten(1)    # [n, 27]


def _check_gated(out, acc, bias, geglu, what):
    """A gated (SwiGLU / GeGLU) output against the fp32 restatement: 3 ulps of |out| * (1 + |gate|) (a 1-ulp flip of the rounded gate
    g moves silu(g) by up to (1 + |g|) ulps: silu'(g) / silu(g) ~ 1 + 1/g for g << 0) is the hard bound; an element beyond it must
    be one rounding flip of gate / activation / up away from the reference (assert_close_bf16_explained), at most 1e-6 of the outputs."""
    M, N = acc.shape
    pre = acc + (bias.float() if bias is not None else 0.0)
    a = pre.view(M, N // 32, 2, 16)
    g_pre, u_pre = a[:, :, 0, :].reshape(M, N // 2), a[:, :, 1, :].reshape(M, N // 2)
    want = _swiglu_ref(acc, bias, geglu=geglu)
    gate = rbf(g_pre)
    return assert_close_bf16_explained(out, want, ulps=3.0, atol=2e-3, what=what, mag=want.abs() * (1.0 + gate.abs()),
                                       candidates=lambda idx: _gated_candidates(g_pre.flatten()[idx], u_pre.flatten()[idx], geglu))


def _swiglu_ref(acc, bias=None, geglu=False):
    # weight (and bias) rows interleaved [16 gate][16 up]; geglu: tanh GELU instead of SiLU on the gate (Gemma)
    M, N = acc.shape
    if bias is not None:
        acc = acc + bias.float()
    a = acc.view(M, N // 32, 2, 16)
    g, u = rbf(a[:, :, 0, :]), rbf(a[:, :, 1, :])
    act = torch.nn.functional.gelu(g, approximate="tanh") if geglu else torch.nn.functional.silu(g)
    return (rbf(act) * u).reshape(M, N // 2)


# shapes with M >= 1024 and N >= 256 run the 256x256 8-wave kernel (gemm256.hip), the others the 128x128 one
WIDE_GEMM_SHAPES = [(128, 128, 64), (300, 200, 128), (1000, 384, 1216), (257, 1536, 1536), (1024, 256, 64), (1300, 512, 128),
                    (2048, 768, 1216), (1111, 1000, 192), (5184, 1280, 320)]
WIDE_GEMM_EPIS = [0, 1, 2, 3, 6]
# The launch geometry of `python bench.py` (12 pages of 1008 x 1008 per tower launch = 62 208 rows, 16 prompts of 1344 padded rows
# per prefill launch = 21 504 rows; engine.wide_plan lists them): every one of these has MORE tiles than the 256 workgroups of the
# persistent grid, so a workgroup walks several tiles — next tile's prologue issued before the epilogue, counted hand-over wait,
# residual rows fetched two passes ahead — which none of the shapes above does for epilogues 0 / 1 / 2 / 3 / 6.
# Then one shape per remaining (epilogue, more than one round) class of the other presets: 7B / PaliGemma widths, K = 3456, 18 944.
WIDE_BENCH_CASES = [
    (62208, 1280, 1216, 0), (62208, 1280, 1280, 1), (62208, 5120, 1280, 2), (62208, 1280, 5120, 1),   # patch embed, proj, fc1, fc2
    (15552, 5120, 5120, 3), (15552, 1536, 5120, 0),                                                    # merger
    (21504, 2048, 1536, 0), (21504, 1536, 1536, 1), (21504, 1536, 8960, 1),                            # prefill qkv, o, down
    (20000, 1280, 1280, 1), (15552, 5120, 1280, 2),          # ragged last row panel (M % 256 != 0) on the multi-tile path
    (49152, 4352, 1152, 6), (49152, 2048, 1152, 0),          # SigLIP fc1 (tanh GELU) and the projector in bf16
    (62208, 1280, 3456, 1), (21504, 3584, 18944, 1),         # Qwen2.5-VL-7B tower down / decoder down: 54 and 296 K tiles
]


def _check_epilogue(out, acc, bias, res, epi, what):
    """2 bf16 ulps of the largest rounded intermediate (bf16(acc+bias), the residual) + fp32 accumulation-order noise is the hard
    bound.  Activation epilogues have three rounding points (bf16(acc + bias), the rounded gate, the output) whose flips can coincide
    in one element (12 of 3.2e8 outputs at 2.1-2.2 ulps on the first GPU run of the bench-geometry cases): such an element must be
    EXPLAINED — equal to the epilogue of the reference accumulator with those intermediates moved by one ulp — and stay below 1e-6
    of the outputs (assert_close_bf16_explained); linear / residual epilogues have no such chain and keep the plain bound."""
    want = _epilogue_ref(acc, bias, res, epi)
    mag = (acc + bias.float()).abs() + (res.float().abs() if epi == 1 else 0.0)
    if epi in (0, 1):
        assert_close_bf16(out, want, ulps=2.0, atol=2e-3, what=what, mag=mag)
        return 0
    v = acc + bias.float()
    return assert_close_bf16_explained(out, want, ulps=2.0, atol=2e-3, what=what, mag=mag,
                                       candidates=lambda idx: _activation_candidates(v.flatten()[idx], epi))


def _check_gemm_wide(M, N, K, epi):
    x = randbf(M, K, scale=1.0, seed=1)
    w = randbf(N, K, scale=K ** -0.5, seed=2)
    bias = randbf(N, scale=0.5, seed=3)
    res = randbf(M, N, seed=4)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    rc = lib().hwocr_gemm_wide(p(x), p(w), p(bias), p(res) if epi == 1 else None, p(out), M, N, K, K, K, N, N, epi, st())
    assert rc == 0
    sync()
    acc = x.float() @ w.float().t()
    return _check_epilogue(out, acc, bias, res, epi, f"gemm_wide epi={epi} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", WIDE_GEMM_SHAPES)
@pytest.mark.parametrize("epi", WIDE_GEMM_EPIS)
def test_gemm_wide(M, N, K, epi):
    _check_gemm_wide(M, N, K, epi)


# The four-wave form of the 256 x 256 kernel (csrc/gemm256w4.hip; taken for plain / gated epilogues at any K and the others up to
# K = 2048): its K loop has three forms - 2 K tiles (no steady state, no DMA inside), 3 (first + the two closing tiles), more - and its
# tile loop two (one tile per workgroup / several, with ragged last panels: the counted wait of the next tile's first K tile then
# takes the conservative branch).  M >= 1024 so that hwocr_gemm_wide takes the 256 x 256 kernel at all.
W4_LOOP_CASES = [(1024, 512, 128, 0), (1024, 512, 192, 0), (1100, 520, 256, 0), (1100, 520, 320, 2), (1024, 512, 128, 1), (2048, 768, 192, 1),
                 (70000, 1032, 128, 1), (70000, 1032, 192, 2), (70000, 1032, 128, 0)]


@pytest.mark.parametrize("M,N,K,epi", W4_LOOP_CASES)
def test_gemm_wide_four_wave_loop_forms(M, N, K, epi):
    _check_gemm_wide(M, N, K, epi)


@pytest.mark.parametrize("M,N,K,epi", WIDE_BENCH_CASES)
def test_gemm_wide_bench_geometry(M, N, K, epi):
    """Every workgroup of the persistent 256 x 256 kernel walks more than one tile (tiles > 256) — against the fp32 product."""
    assert ((M + 255) // 256) * ((N + 255) // 256) > 256
    # (activation epilogues among 3e8 outputs: a handful of elements land at 2.1-2.2 ulps — three rounding flips coinciding;
    # _check_epilogue holds each of them to that explanation instead of widening the bound)
    _check_gemm_wide(M, N, K, epi)


# with_bias: the Qwen2.5-VL vision MLP (gate_proj / up_proj carry a bias); (5184, 6912, 1280) is its page shape
# (21504, 17920, 1536): the bench's prefill gate/up launch (5880 tiles, 23 per workgroup)
WIDE_SWIGLU_SHAPES = [(200, 256, 128), (129, 17920 // 10, 1536), (1328, 1792, 1536), (2100, 608, 192), (5184, 6912, 1280),
                      (21504, 17920, 1536)]
WIDE_GEGLU_SHAPES = [(200, 256, 128), (1328, 1792, 1536), (4096, 8192, 2048)]   # the last: 512 tiles (Gemma's GeGLU over several rounds)


@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("M,N,K", WIDE_SWIGLU_SHAPES)
def test_gemm_wide_swiglu(M, N, K, with_bias):
    N = (N // 32) * 32
    x = randbf(M, K, seed=5)
    w = randbf(N, K, scale=K ** -0.5, seed=6)
    bias = randbf(N, scale=0.5, seed=7) if with_bias else None
    out = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
    rc = lib().hwocr_gemm_wide(p(x), p(w), p(bias) if with_bias else None, None, p(out), M, N, K, K, K, N // 2, 0, 4, st())
    assert rc == 0
    sync()
    _check_gated(out, x.float() @ w.float().t(), bias, False, f"gemm_wide swiglu {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", WIDE_GEGLU_SHAPES)
def test_gemm_wide_geglu(M, N, K):
    x = randbf(M, K, seed=5)
    w = randbf(N, K, scale=K ** -0.5, seed=6)
    out = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_gemm_wide(p(x), p(w), None, None, p(out), M, N, K, K, K, N // 2, 0, 7, st()) == 0
    sync()
    _check_gated(out, x.float() @ w.float().t(), None, True, f"gemm_wide geglu {M}x{N}x{K}")


def test_gemm_wide_rejects_bad_shapes():
    x = randbf(64, 72)
    assert lib().hwocr_gemm_wide(p(x), p(x), None, None, p(x), 64, 64, 72, 72, 72, 64, 0, 0, st()) == 1  # K % 64


def _tiled(w):
    n, k = w.shape
    out = torch.empty(n * k, dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_tile_weights(p(w), p(out), n, k, k, st()) == 0
    return out


def test_tile_weights_layout():
    n, k = 48, 96
    w = (torch.arange(n * k, dtype=torch.float32) % 251).view(n, k).to(torch.bfloat16).to(DEV)
    t = _tiled(w).view(n // 16, k // 32, 4, 16, 8).cpu()
    want = w.cpu().view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4)
    assert torch.equal(t, want)


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("B", [1, 7, 16, 48, 96, 128, 190, 256])
@pytest.mark.parametrize("N,K", [(96, 64), (2048, 1536), (1536, 2304)])
def test_gemm_skinny_linear_and_partial(B, N, K, tiled):
    x = randbf(B, K, seed=7)
    w = randbf(N, K, scale=K ** -0.5, seed=8)
    wk = _tiled(w) if tiled else w
    bias = randbf(N, scale=0.5, seed=9)
    acc = x.float() @ w.float().t()
    out = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_gemm_skinny(p(x), p(wk), p(bias), p(out), B, N, K, K, K, N, 0, 1, tiled, st()) == 0
    sync()
    assert_close_bf16(out, acc + bias.float(), ulps=2.0, atol=2e-3, what="skinny linear")
    chunks = (K + 255) // 256
    for splitk in sorted({1, min(3, chunks), chunks}):
        per = (chunks + splitk - 1) // splitk
        if (splitk - 1) * per * 256 >= K:
            continue
        slabs = torch.full((splitk, B, N), float("nan"), dtype=torch.float32, device=DEV)
        assert lib().hwocr_gemm_skinny(p(x), p(wk), None, p(slabs), B, N, K, K, K, N, 5, splitk, tiled, st()) == 0
        sync()
        got = slabs.sum(0)
        assert torch.allclose(got, acc, rtol=1e-4, atol=2e-3), f"partial splitk={splitk}: {(got-acc).abs().max()}"


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("B", [3, 48, 96, 252])
def test_gemm_skinny_swiglu(B, tiled):
    N, K = 2 * 1792, 1536
    x = randbf(B, K, seed=10)
    w = randbf(N, K, scale=K ** -0.5, seed=11)
    wk = _tiled(w) if tiled else w
    for epi, geglu in ((4, False), (7, True)):  # SwiGLU (Qwen), GeGLU (Gemma)
        out = torch.full((B, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_skinny(p(x), p(wk), None, p(out), B, N, K, K, K, N // 2, epi, 1, tiled, st()) == 0
        sync()
        assert_close_bf16(out, _swiglu_ref(x.float() @ w.float().t(), geglu=geglu), ulps=3.0, atol=2e-3,
                          what=f"skinny glu epi={epi}")


# The decode-step GEMMs exactly as hwocr_decode_step issues them for the shipped presets at the bench's read counts
# (126 = the pre-r01g default, 252 = `python bench.py`): (N, K, epi, splitk) from engine.decode_plan — qkv / o / gate-up /
# down / LM head of Qwen2-VL-2B, Qwen2.5-VL-7B (olmOCR-2), Qwen2.5-VL-3B, PaliGemma-3B (GeGLU) and the `small` preset.
# tests/test_decode_variants.py (CPU) walks every preset x read count through the library's launch planner and fails if a
# kernel instance it picks has no case here.
DECODE_GEMM_SHAPES = {
    126: [(2048, 1536, 5, 4), (1536, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 12), (151936, 1536, 0, 1),
          (4608, 3584, 5, 8), (3584, 3584, 5, 8), (37888, 3584, 4, 1), (3584, 18944, 5, 12), (152064, 3584, 0, 1),
          (2560, 2048, 5, 5), (2048, 2048, 5, 5), (22016, 2048, 4, 1), (2048, 11008, 5, 12), (151936, 2048, 0, 1),
          (32768, 2048, 7, 1), (2048, 16384, 5, 12), (257216, 2048, 0, 1),
          (1280, 768, 5, 2), (768, 768, 5, 2), (6144, 768, 4, 1), (768, 3072, 5, 8), (32768, 768, 0, 1)],
    252: [(2048, 1536, 5, 4), (1536, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 8), (151936, 1536, 0, 1),
          (4608, 3584, 5, 4), (3584, 3584, 5, 4), (37888, 3584, 4, 1), (3584, 18944, 5, 8), (152064, 3584, 0, 1),
          (2560, 2048, 5, 4), (2048, 2048, 5, 4), (22016, 2048, 4, 1), (2048, 11008, 5, 8), (151936, 2048, 0, 1),
          (32768, 2048, 7, 1), (2048, 16384, 5, 8), (257216, 2048, 0, 1),
          (1280, 768, 5, 2), (768, 768, 5, 2), (6144, 768, 4, 1), (768, 3072, 5, 8), (32768, 768, 0, 1)],
    # one page = 3 reads in flight (BASELINE config 2 as literally stated), and the 129..256-row kernels away from 252
    3: [(2048, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 12), (151936, 1536, 0, 1), (32768, 2048, 7, 1)],
    130: [(2048, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 8), (151936, 1536, 0, 1), (37888, 3584, 4, 1)],
    256: [(17920, 1536, 4, 1), (1536, 8960, 5, 8), (151936, 1536, 0, 1), (32768, 2048, 7, 1)],
    # 17..32 reads (the two-row-tile instance; <= 16 reads take csrc/gemm_rows16.hip: ROWS16_CASES): the tail of a continuous batch
    24: [(2048, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 12), (151936, 1536, 0, 1), (32768, 2048, 7, 1), (32768, 768, 0, 1)],
}
DECODE_GEMM_CASES = [(B,) + shape for B, shapes in DECODE_GEMM_SHAPES.items() for shape in shapes]


@pytest.mark.parametrize("B,N,K,epi,splitk", DECODE_GEMM_CASES)
def test_gemm_skinny_decode_shapes(B, N, K, epi, splitk):
    """hwocr_gemm_skinny(tiled weights) on the shapes, epilogues and split-K the decode step uses, at its row counts,
    against the fp32 product of the same bf16 operands (what HF's nn.Linear + SiLU/GELU-gated MLP compute:
    modeling_qwen2_vl.py:453-466, :501-504; gemma/modeling_gemma.py:84-97)."""
    x = randbf(B, K, seed=70)
    w = randbf(N, K, scale=K ** -0.5, seed=71)
    wk = _tiled(w)
    acc = x.float() @ w.float().t()
    if epi == 5:
        slabs = torch.full((splitk, B, N), float("nan"), dtype=torch.float32, device=DEV)
        assert lib().hwocr_gemm_skinny(p(x), p(wk), None, p(slabs), B, N, K, K, K, N, 5, splitk, 1, st()) == 0
        sync()
        got = slabs.sum(0)
        assert torch.isfinite(got).all(), "a slab element was left unwritten"
        assert torch.allclose(got, acc, rtol=1e-4, atol=2e-3), f"partial splitk={splitk}: {(got - acc).abs().max()}"
    elif epi == 0:
        out = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_skinny(p(x), p(wk), None, p(out), B, N, K, K, K, N, 0, 1, 1, st()) == 0
        sync()
        assert_close_bf16(out, acc, ulps=2.0, atol=2e-3, what="decode linear (LM head)")
    else:
        out = torch.full((B, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_skinny(p(x), p(wk), None, p(out), B, N, K, K, K, N // 2, epi, 1, 1, st()) == 0
        sync()
        want = _swiglu_ref(acc, geglu=(epi == 7))
        gate = rbf(acc.view(B, N // 32, 2, 16)[:, :, 0, :]).reshape(B, N // 2)
        assert_close_bf16(out, want, ulps=3.0, atol=2e-3, what=f"decode glu epi={epi}", mag=want.abs() * (1.0 + gate.abs()))


# ------------------------------------------------------------------------------------------------ attention
# ---- the decode GEMMs with at most 16 reads in flight (csrc/gemm_rows16.hip): (rows, N, K, epi, splitk, norm prologue?, slabs summed
# by the prologue, gemma) as hwocr_decode_step issues them per preset — qkv (norm, the previous layer's down slabs), o (residual in
# place), gate/up (norm, gated), down (split-K slabs) — at 1 / 3 / 7 / 16 rows; tests/test_decode_variants.py holds every preset's
# plan against the classes of this list
ROWS16_CASES = [
    # Qwen2-VL-2B
    (3, 2048, 1536, 5, 1, True, 0, 0), (3, 2048, 1536, 5, 1, True, 2, 0), (3, 1536, 1536, 1, 1, False, 0, 0),
    (3, 17920, 1536, 4, 1, True, 0, 0), (3, 1536, 8960, 5, 2, False, 0, 0),
    (1, 2048, 1536, 5, 1, True, 2, 0), (16, 2048, 1536, 5, 1, True, 2, 0), (7, 17920, 1536, 4, 1, True, 0, 0), (16, 1536, 8960, 5, 2, False, 0, 0),
    (16, 1536, 1536, 1, 1, False, 0, 0), (1, 17920, 1536, 4, 1, True, 0, 0),
    # Qwen2.5-VL-7B / -3B
    (3, 4608, 3584, 5, 1, True, 1, 0), (3, 3584, 3584, 1, 1, False, 0, 0), (3, 37888, 3584, 4, 1, True, 0, 0), (3, 3584, 18944, 5, 1, False, 0, 0),
    (3, 2560, 2048, 5, 1, True, 2, 0), (3, 22016, 2048, 4, 1, True, 0, 0), (3, 2048, 11008, 5, 2, False, 0, 0),
    # PaliGemma-3B (Gemma norm, GeGLU)
    (3, 2560, 2048, 5, 1, True, 2, 1), (3, 2048, 2048, 1, 1, False, 0, 0), (3, 32768, 2048, 7, 1, True, 0, 1), (3, 2048, 16384, 5, 2, False, 0, 0),
    # `small` preset (4 slabs: the most a prologue sums) and the tiny golden models (K = 256: fewer k-steps than K slices)
    (3, 1280, 768, 5, 1, True, 4, 0), (3, 768, 768, 1, 1, False, 0, 0), (3, 6144, 768, 4, 1, True, 0, 0), (3, 768, 3072, 5, 4, False, 0, 0),
    (2, 512, 256, 5, 1, True, 1, 0), (2, 256, 256, 1, 1, False, 0, 0), (2, 512, 256, 4, 1, True, 0, 0), (2, 256, 256, 5, 1, False, 0, 0),
    (2, 768, 256, 5, 1, True, 1, 1), (2, 256, 512, 1, 1, False, 0, 0), (2, 1024, 256, 7, 1, True, 0, 1), (2, 256, 512, 5, 1, False, 0, 0),
]


def _rows16_norm_ref(h, slabs, w, gemma, eps=1e-6):
    """(h', x): the residual update and the normalised GEMM input of the prologue, reference's rounding chain."""
    hf = h.float()
    if slabs is not None:
        y = torch.zeros_like(hf)
        for sl in slabs:           # ascending slab order, as the kernel
            y = y + sl
        hf = rbf(rbf(y) + hf)
    xhat = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)
    x = rbf(xhat * (1.0 + w.float())) if gemma else rbf(w.float() * rbf(xhat))
    return hf, x


@pytest.mark.parametrize("B,N,K,epi,splitk,norm,nslab,gemma", ROWS16_CASES)
def test_gemm_rows16(B, N, K, epi, splitk, norm, nslab, gemma):
    """hwocr_gemm_rows16 against the fp32 product of the same bf16 operands, with the RMSNorm prologue (residual update written back
    once, bit-exact; normalised rows as hwocr_add_rmsnorm) and every epilogue the <= 16-read decode layer uses."""
    import ctypes as C

    from handwritten_ocr_amd import _lib
    w = randbf(N, K, scale=K ** -0.5, seed=91)
    wk = _tiled(w)
    if norm:
        h = randbf(B, K, scale=2.0, seed=92)
        nw = randbf(K, scale=0.3, seed=93) + (0.0 if gemma else 1.0)
        slabs = randf32(max(nslab, 1), B, K, seed=901) if nslab else None
        h_out = torch.full((B, K), float("nan"), dtype=torch.bfloat16, device=DEV)
        blk = _lib.Rows16Norm(h_in=p(h), h_out=p(h_out), ldh=K, slabs=p(slabs) if nslab else None, nslab=nslab, slab_stride=B * K, ld_slab=K,
                              norm_w=p(nw), eps=1e-6, gemma=gemma)
        hp, x = _rows16_norm_ref(h, slabs[:nslab] if nslab else None, nw, gemma)
        nref, xptr, ldx = C.byref(blk), None, 0
    else:
        xb = randbf(B, K, seed=94)
        x, nref, xptr, ldx = xb.float(), None, p(xb), K
    acc = x @ w.float().t()
    # a normalised element that rounds to the neighbouring bf16 (the row statistic is summed in another order) moves a sum of K terms
    # by |w| ulp(x): far below the output's own rounding; the fp32 slabs get an absolute allowance for it
    if epi == 5:
        out = torch.full((splitk, B, N), float("nan"), dtype=torch.float32, device=DEV)
        assert lib().hwocr_gemm_rows16(xptr, ldx, p(wk), p(out), N, B, N, K, 5, splitk, nref, st()) == 0
        sync()
        assert torch.isfinite(out).all(), "a slab element was left unwritten"
        got = out.sum(0)
        assert torch.allclose(got, acc, rtol=1e-4, atol=6e-3 if norm else 2e-3), f"partial splitk={splitk}: {(got - acc).abs().max()}"
    elif epi == 1:
        res = randbf(B, N, seed=95)
        out = res.clone()
        assert lib().hwocr_gemm_rows16(xptr, ldx, p(wk), p(out), N, B, N, K, 1, 1, nref, st()) == 0
        sync()
        assert_close_bf16(out, rbf(acc) + res.float(), ulps=2.0, atol=2e-3, what="rows16 residual", mag=acc.abs() + res.float().abs())
    else:
        out = torch.full((B, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_rows16(xptr, ldx, p(wk), p(out), N // 2, B, N, K, epi, 1, nref, st()) == 0
        sync()
        want = _swiglu_ref(acc, geglu=(epi == 7))
        gate = rbf(acc.view(B, N // 32, 2, 16)[:, :, 0, :]).reshape(B, N // 2)
        assert_close_bf16(out, want, ulps=3.0, atol=3e-3, what="rows16 gated", mag=want.abs() * (1.0 + gate.abs()))
    if norm:
        assert torch.equal(h_out.float(), hp), "the residual update written back by the prologue"


def test_gemm_rows16_rejects_what_it_cannot_run():
    x = randbf(16, 64)
    assert lib().hwocr_gemm_rows16(p(x), 64, p(x), p(x), 64, 17, 64, 64, 0, 1, None, st()) == 1    # more than 16 rows
    assert lib().hwocr_gemm_rows16(p(x), 64, p(x), p(x), 64, 4, 64, 64, 1, 2, None, st()) == 1     # split-K without PARTIAL
    assert lib().hwocr_gemm_rows16(None, 0, p(x), p(x), 64, 4, 64, 64, 0, 1, None, st()) == 1      # neither rows nor a norm block


def _sdpa_ref(q, k, v, causal, scale):
    # q [Hq, L, d], k/v [Hkv, L, d] (fp32); returns [L, Hq, d]
    Hq, Lq, d = q.shape
    g = Hq // k.shape[0]
    k = k.repeat_interleave(g, 0)
    v = v.repeat_interleave(g, 0)
    s = (q @ k.transpose(1, 2)) * scale
    if causal:
        s = s + torch.full((Lq, Lq), float("-inf"), device=q.device).triu(1)
    return (torch.softmax(s, -1) @ v).permute(1, 0, 2)


# (64, 6, 6, False): the `small` preset's tower heads; (128, 28, 4, True, tiled): the 7B decoder's grouping
ATTN_PREFILL_CASES = [(80, 4, 4, False, 0), (128, 6, 2, True, 0), (32, 2, 2, False, 0), (64, 2, 1, True, 0), (128, 2, 2, False, 0),
                      (128, 6, 2, True, 1), (256, 4, 1, False, 0), (256, 2, 1, True, 0), (64, 6, 6, False, 0), (128, 28, 4, True, 1)]


@pytest.mark.parametrize("hd,Hq,Hkv,causal,tiled", ATTN_PREFILL_CASES)
def test_attn_prefill(hd, Hq, Hkv, causal, tiled):
    lens = [300, 64, 37, 129]
    nseg, Lp = len(lens), 320  # per-segment stride, multiple of 64
    q = randbf(nseg, Lp, Hq, hd, seed=12)
    k = randbf(nseg, Hkv, Lp, hd, seed=13)
    v = randbf(nseg, Hkv, Lp, hd, seed=14)
    vt = v.transpose(2, 3).contiguous()  # [nseg][Hkv][hd][Lp]
    # poison the key padding: it must never reach the output
    for s, n in enumerate(lens):
        vt[s, :, :, n:] = float("nan")
        k[s, :, n:, :] = 1e4
    out = torch.zeros(nseg, Lp, Hq * hd, dtype=torch.bfloat16, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    scale = hd ** -0.5
    kk, vv = (tile_k(k), tile_v(vt)) if tiled else (k, vt)
    rc = lib().hwocr_attn_prefill(p(q), p(kk), p(vv), p(out), p(lens_d), nseg, Hq, Hq // Hkv, hd, max(lens), int(causal),
                                  Lp * Hq * hd, hd, Hq * hd, Hkv * Lp * hd, Lp * hd, hd,
                                  Hkv * hd * Lp, hd * Lp, Lp, Lp * Hq * hd, Hq * hd, scale, tiled, st())
    assert rc == 0
    sync()
    for s, n in enumerate(lens):
        want = _sdpa_ref(q[s, :n].float().permute(1, 0, 2), k[s, :, :n].float(), v[s, :, :n].float(), causal, scale)
        got = out[s, :n].view(n, Hq, hd)
        # P is rounded to bf16 before the PV product (as the reference's bf16 attention does): ~2^-8 relative on
        # each of the weights -> allow 4 output ulps + 4e-3 absolute
        assert_close_bf16(got, want, ulps=4.0, atol=4e-3, what=f"attn_prefill seg {s}")
    assert torch.isfinite(out.float()).all()


ATTN_VIT80_LONG_LENS = [[2000, 1537, 383, 769], [2048, 64, 1, 1600], [255, 257, 1999, 130]]
ATTN_VARLEN_CASES = [(80, 4), (32, 2)]
ATTN_HD256_LENS = [700, 513, 64, 1, 640, 333, 65, 128]
MROPE_CASES = [(128, 16, 40, 0), (128, 16, 40, 1), (256, 128, 128, 0)]


@pytest.mark.parametrize("kernel", ["x", "12", "4"])
@pytest.mark.parametrize("lens", ATTN_VIT80_LONG_LENS)
def test_attn_vit80_long_segments(kernel, lens, monkeypatch):
    """Page-length segments take the one-wave-per-SIMD form of the head_dim-80 kernel (attention_vit80x.hip: 256 queries per
    workgroup, output accumulators in asm-owned AGPRs); the 12-wave (384 queries) and 4-wave (128) forms stay selectable
    (HWOCR_VIT80_KERNEL, read per call).  Ragged lengths around the block sizes, whole and partial last key tiles, segments of
    one tile and of one token, K rows / V^T columns past each segment poisoned."""
    monkeypatch.setenv("HWOCR_VIT80_KERNEL", kernel)
    hd, heads = 80, 2
    nseg, Lp = len(lens), 2048
    q = randbf(nseg, heads, Lp, hd, seed=41)
    k = randbf(nseg, heads, Lp, hd, seed=42)
    v = randbf(nseg, heads, Lp, hd, seed=43)
    vt = v.transpose(2, 3).contiguous()
    for s_, n in enumerate(lens):
        vt[s_, :, :, n:] = float("nan")
        k[s_, :, n:, :] = 1e4
    out = torch.zeros(nseg, Lp, heads * hd, dtype=torch.bfloat16, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    scale = hd ** -0.5
    rc = lib().hwocr_attn_prefill(p(q), p(k), p(vt), p(out), p(lens_d), nseg, heads, 1, hd, max(lens), 0,
                                  heads * Lp * hd, Lp * hd, hd, heads * Lp * hd, Lp * hd, hd,
                                  heads * hd * Lp, hd * Lp, Lp, Lp * heads * hd, heads * hd, scale, 0, st())
    assert rc == 0
    sync()
    for s_, n in enumerate(lens):
        want = _sdpa_ref(q[s_, :, :n].float(), k[s_, :, :n].float(), v[s_, :, :n].float(), False, scale)
        assert_close_bf16(out[s_, :n].view(n, heads, hd), want, ulps=4.0, atol=4e-3, what=f"attn_vit80 kernel {kernel} seg {s_}")
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("kernel", ["x", "12"])
def test_attn_vit80_page_shape(kernel, monkeypatch):
    """The bench's page: 5184 tokens (72 x 72 patches of a 1008 x 1008 page), the tower's buffer layout ([heads][rows][80] q / k,
    [heads][80][rows] v^T, two pages on one row axis), scores with a spread of maxima so that the lazy reference moves on some
    queries late in the sweep; against fp32 SDPA, and bit-reproducible run to run."""
    monkeypatch.setenv("HWOCR_VIT80_KERNEL", kernel)
    hd, heads, P, nimg = 80, 2, 5184, 2
    rows, DH = nimg * P, heads * hd
    q = randbf(heads, rows, hd, seed=51)
    k = randbf(heads, rows, hd, seed=52)
    k[:, P - 300:P - 290] *= 6.0          # a few late keys with large scores: the running reference moves after 4800 keys
    v = randbf(heads, rows, hd, seed=53)
    vt = torch.zeros(heads, hd, rows + 64, dtype=torch.bfloat16, device=DEV)[:, :, :rows]
    vt.copy_(v.transpose(1, 2))
    lens = torch.full((nimg,), P, dtype=torch.int32, device=DEV)
    scale = hd ** -0.5
    outs = []
    for _ in range(2):
        out = torch.zeros(rows, DH, dtype=torch.bfloat16, device=DEV)
        rc = lib().hwocr_attn_prefill(p(q), p(k), p(vt), p(out), p(lens), nimg, heads, 1, hd, P, 0, P * hd, rows * hd, hd, P * hd, rows * hd,
                                      hd, P, hd * (rows + 64), rows + 64, P * DH, DH, scale, 0, st())
        assert rc == 0
        sync()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    for i in range(nimg):
        sl = slice(i * P, (i + 1) * P)
        want = _sdpa_ref(q[:, sl].float(), k[:, sl].float(), v[:, sl].float(), False, scale)
        assert_close_bf16(outs[0][sl].view(P, heads, hd), want, ulps=4.0, atol=4e-3, what=f"page {i}, kernel {kernel}")


@pytest.mark.parametrize("hd,heads", ATTN_VARLEN_CASES)
def test_attn_varlen_windows(hd, heads):
    """Ragged windows packed on one row axis (Qwen2.5-VL windowed layers): starts are multiples of 4 rows only."""
    lens = [64, 16, 32, 4, 64, 36, 8, 48, 12, 64]
    offs = [0]
    for n in lens[:-1]:
        offs.append(offs[-1] + n)
    rows = 384  # buffer rows (multiple of 64) >= sum(lens) = 348; rows past the last window are padding
    assert offs[-1] + lens[-1] <= rows
    q = randbf(heads, rows, hd, seed=31)
    k = randbf(heads, rows, hd, seed=32)
    v = randbf(heads, rows, hd, seed=33)
    vt = torch.zeros(heads * hd * rows + 64, dtype=torch.bfloat16, device=DEV)  # 64 elements of slack, as the ABI asks
    vt[: heads * hd * rows] = v.transpose(1, 2).reshape(-1)
    out = torch.zeros(rows, heads * hd, dtype=torch.bfloat16, device=DEV)
    off_d = torch.tensor(offs, dtype=torch.int32, device=DEV)
    len_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    scale = hd ** -0.5
    rc = lib().hwocr_attn_varlen(p(q), p(k), p(vt), p(out), p(off_d), p(len_d), len(lens), heads, hd, max(lens),
                                 rows * hd, hd, rows * hd, hd, hd * rows, rows, heads * hd, scale, st())
    assert rc == 0
    sync()
    for o, n in zip(offs, lens):
        want = _sdpa_ref(q[:, o: o + n].float(), k[:, o: o + n].float(), v[:, o: o + n].float(), False, scale)
        got = out[o: o + n].view(n, heads, hd)
        assert_close_bf16(got, want, ulps=4.0, atol=4e-3, what=f"attn_varlen window at {o}")
    assert (out[offs[-1] + lens[-1]:] == 0).all(), "rows outside every window must stay untouched"


def _same_with_the_last_workgroup_merging(out, ptrs, tail, part_o, part_ml):
    """The split form again with arrival counters (hwocr.h: the workgroup of a (read, kv head) that arrives last merges the partials,
    no merge launch): the bytes of the two-launch form, twice in a row (the counters go back to zero), counters zero afterwards."""
    B, Hq, Hkv, nsplit = tail[:4]
    if nsplit == 1:
        return
    arrive = torch.zeros(B * Hkv, dtype=torch.int32, device=DEV)
    for _ in range(2):
        part_o.fill_(float("nan"))
        part_ml.fill_(float("nan"))
        got = torch.full_like(out, float("nan"))
        assert lib().hwocr_attn_decode(*ptrs, p(got), p(part_o), p(part_ml), p(arrive), *tail, st()) == 0
        sync()
        assert torch.equal(got.view(torch.int16), out.view(torch.int16)), "last-workgroup merge differs from the merge launch"
        assert int(arrive.abs().sum()) == 0


# hd 256: Gemma (MQA: 8 query heads on one kv head), row layout only
@pytest.mark.parametrize("hd,Hq,Hkv,tiled", [(128, 12, 2, 0), (128, 12, 2, 1), (256, 8, 1, 0)])
@pytest.mark.parametrize("nsplit", [1, 4])
def test_attn_decode(nsplit, hd, Hq, Hkv, tiled):
    ctx = 640
    lens = [1, 63, 64, 65, 500, 640]
    B = len(lens)
    q = randbf(B, Hq, hd, seed=15)
    k = randbf(B, Hkv, ctx, hd, seed=16)
    v = randbf(B, Hkv, ctx, hd, seed=17)
    vt = v.transpose(2, 3).contiguous()
    for b, n in enumerate(lens):
        vt[b, :, :, n:] = float("nan")
        k[b, :, n:, :] = 1e4
    out = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    G = Hq // Hkv
    part_o = torch.zeros(B * Hkv * nsplit * G * hd, dtype=torch.float32, device=DEV)
    part_ml = torch.zeros(B * Hkv * nsplit * G * 2, dtype=torch.float32, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    kk, vv = (tile_k(k), tile_v(vt)) if tiled else (k, vt)
    rc = lib().hwocr_attn_decode(p(q), p(kk), p(vv), p(lens_d), p(out), p(part_o), p(part_ml), None, B, Hq, Hkv, nsplit,
                                 Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx, hd ** -0.5, hd, tiled, st())
    assert rc == 0
    sync()
    _same_with_the_last_workgroup_merging(out, (p(q), p(kk), p(vv), p(lens_d)), (B, Hq, Hkv, nsplit, Hkv * ctx * hd, ctx * hd,
                                          Hkv * hd * ctx, hd * ctx, ctx, hd ** -0.5, hd, tiled), part_o, part_ml)
    for b, n in enumerate(lens):
        qq = q[b].float().unsqueeze(1)  # [Hq,1,d]
        want = _sdpa_ref(qq, k[b, :, :n].float(), v[b, :, :n].float(), False, hd ** -0.5).reshape(Hq * hd)
        assert_close_bf16(out[b], want, ulps=4.0, atol=4e-3, what=f"attn_decode read {b} len {n}")


# (B, Hq, Hkv, hd, tiled, ctx, nsplit): the decode attention as the bench runs it — 252 / 126 reads x 2048 cached positions,
# lengths spread over the 512 generated tokens past the 1328-token prompt (Qwen2-VL-2B, Qwen2.5-VL-7B head counts; the row
# layout with 256-wide heads for Gemma at 126 reads = 4 splits + merge and at 252 = one pass)
ATTN_DECODE_BENCH_CASES = [(252, 12, 2, 128, 1, 2048, 1), (126, 12, 2, 128, 1, 2048, 1), (252, 28, 4, 128, 1, 2048, 1),
                           (3, 12, 2, 128, 1, 2048, 16), (126, 8, 1, 256, 0, 1280, 6), (252, 8, 1, 256, 0, 1280, 1)]


@pytest.mark.parametrize("B,Hq,Hkv,hd,tiled,ctx,nsplit", ATTN_DECODE_BENCH_CASES)
def test_attn_decode_bench_shapes(B, Hq, Hkv, hd, tiled, ctx, nsplit):
    g = torch.Generator().manual_seed(5)
    lo = ctx * 1328 // 2048
    lens = torch.randint(lo, ctx - 200, (B,), generator=g).tolist()
    lens[0], lens[-1] = lo, ctx  # the shortest and the full cache
    q = randbf(B, Hq, hd, seed=15)
    k = randbf(B, Hkv, ctx, hd, seed=16)
    v = randbf(B, Hkv, ctx, hd, seed=17)
    vt = v.transpose(2, 3).contiguous()
    for b, n in enumerate(lens):
        vt[b, :, :, n:] = float("nan")
        k[b, :, n:, :] = 1e4
    out = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    G = Hq // Hkv
    part_o = torch.zeros(B * Hkv * nsplit * G * hd, dtype=torch.float32, device=DEV)
    part_ml = torch.zeros(B * Hkv * nsplit * G * 2, dtype=torch.float32, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    kk, vv = (tile_k(k), tile_v(vt)) if tiled else (k, vt)
    rc = lib().hwocr_attn_decode(p(q), p(kk), p(vv), p(lens_d), p(out), p(part_o), p(part_ml), None, B, Hq, Hkv, nsplit,
                                 Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx, hd ** -0.5, hd, tiled, st())
    assert rc == 0
    sync()
    _same_with_the_last_workgroup_merging(out, (p(q), p(kk), p(vv), p(lens_d)), (B, Hq, Hkv, nsplit, Hkv * ctx * hd, ctx * hd,
                                          Hkv * hd * ctx, hd * ctx, ctx, hd ** -0.5, hd, tiled), part_o, part_ml)
    # fp32 reference of all reads at once: scores masked past each read's length
    kf, vf = k.float().repeat_interleave(G, 1), v.float().repeat_interleave(G, 1)
    s = torch.einsum("bhd,bhkd->bhk", q.float(), kf) * hd ** -0.5
    mask = torch.arange(ctx, device=DEV)[None, None, :] >= lens_d[:, None, None]
    pr = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
    want = torch.einsum("bhk,bhkd->bhd", pr, torch.nan_to_num(vf)).reshape(B, Hq * hd)
    assert_close_bf16(out, want, ulps=4.0, atol=4e-3, what="attn_decode at the bench's size")


def test_attn_decode_rejects_more_than_16_splits():
    q = randbf(1, 2, 128)
    assert lib().hwocr_attn_decode(p(q), p(q), p(q), p(q), p(q), p(q), p(q), None, 1, 2, 1, 17, 128 * 64, 128 * 64, 128 * 64,
                                   128 * 64, 64, 1.0, 128, 0, st()) == 1


# ------------------------------------------------------------------------------------------------ row-wise kernels
# (20 000 / 62 208 x 1280): two and four trips of the grid-stride loop (the grid is capped at 4096 workgroups = 16 384 rows; 62 208
# rows is the bench's tower launch); 768 / 576 wide: the two-chunk instance (the `small` preset's decoder width, the tiny SigLIP);
# (70 000 x 768): that instance on its second trip; 3584: the eight-chunk instance
LAYERNORM_CASES = [(5, 64), (1000, 1280), (33, 2048), (20000, 1280), (62208, 1280), (5000, 768), (300, 576), (70000, 768), (49152, 1152),
                   (600, 3584)]


@pytest.mark.parametrize("rows,D", LAYERNORM_CASES)
def test_layernorm(rows, D):
    x = randbf(rows, D, scale=3.0, seed=18)
    w = randbf(D, seed=19)
    b = randbf(D, seed=20)
    out = torch.empty_like(x)
    assert lib().hwocr_layernorm(p(x), p(w), p(b), p(out), rows, D, D, D, 1e-6, st()) == 0
    sync()
    want = torch.nn.functional.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-6)
    assert_close_bf16(out, want, ulps=1.0, atol=1e-3, what="layernorm")
    if rows > 16384:  # a later trip of the loop must read ITS rows: row r of the second trip against row r of a launch of that slice alone
        r0 = 16384 + 5
        part = torch.empty(64, D, dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_layernorm(p(x[r0:]), p(w), p(b), p(part), 64, D, D, D, 1e-6, st()) == 0
        sync()
        assert torch.equal(part, out[r0: r0 + 64])


# D = 3584: Qwen2.5-VL-7B hidden size (512-thread row kernel / 7 chunks per lane); 1280 x 600 rows: its vision tower
# rows > 512: the wave-per-row kernel the tower (Qwen2.5-VL) and the prefill use — 768 / 1280 + 1536 / 2048 / 3584 wide = its 2 / 3 / 4 / 8
# chunk instances; (21504, 1536) and (62208, 1280) are the bench's prefill and 7B-tower launches
ADD_RMSNORM_CASES = [(7, 1536, 0), (96, 1536, 6), (3, 256, 2), (126, 3584, 4), (600, 3584, 0), (600, 1280, 0), (5, 4096, 3),
                     (700, 768, 0), (700, 2048, 0), (21504, 1536, 0), (62208, 1280, 0), (600, 128, 0), (600, 1536, 3)]


@pytest.mark.parametrize("gemma", [0, 1])
@pytest.mark.parametrize("rows,D,nslab", ADD_RMSNORM_CASES)
def test_add_rmsnorm(rows, D, nslab, gemma):
    h = randbf(rows, D, scale=2.0, seed=21)
    w = randbf(D, seed=22)
    bias = randbf(D, seed=23)
    slabs = randf32(max(nslab, 1), rows, D, seed=902)
    h_in = h.clone()
    out = torch.empty_like(h)
    rc = lib().hwocr_add_rmsnorm(p(slabs) if nslab else None, nslab, rows * D, D, p(bias) if nslab else None, p(h), D,
                                 p(w), p(out), D, None, rows, D, 1e-6, gemma, st())
    assert rc == 0
    sync()
    x = h_in.float()
    if nslab:
        y = slabs[:nslab].sum(0) + bias.float()
        x = rbf(rbf(y) + x)
        # h <- bf16(bf16(sum of slabs + bias) + h): the kernel sums the slabs in another order than torch, so its bf16(y) may be the
        # NEIGHBOUR of the reference's where y sits on a rounding boundary; 1 ulp is the hard bound, an element beyond it must be
        # exactly bf16(y' + h) for a neighbour y' of bf16(y) (one in ~1e6 elements: the explained-outlier rule of the GEMM epilogues)
        hin = h_in.float()
        assert_close_bf16_explained(h, x, ulps=1.0, atol=1e-3, what="residual write-back", mag=y.abs() + hin.abs(),
                                    candidates=lambda idx: rbf(bf16_neighbours(rbf(y.flatten()[idx])) + hin.flatten()[idx].unsqueeze(-1)),
                                    max_frac=2e-5)
        x = h.float()
    xhat = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    # Qwen2VLRMSNorm: w * bf16(xhat); GemmaRMSNorm (HF gemma/modeling_gemma.py:70-78): bf16(xhat * (1 + w)), all in fp32
    want = xhat * (1.0 + w.float()) if gemma else w.float() * rbf(xhat)
    # the row statistic is summed in a different order than torch's -> the normalised value may round to the neighbouring
    # bf16 (1 ulp of it, up to 2 ulps of the product across a binade edge) before the weight multiply
    assert_close_bf16(out, want, ulps=2.5, atol=1e-3, what="rmsnorm")


# the final norm of a prefill chunk: last prompt row of every read gathered (2B / `small` widths, the 7B's 512-thread form, Gemma)
ADD_RMSNORM_GATHER_CASES = [(1536, 0), (768, 0), (3584, 0), (2048, 1), (256, 1)]


@pytest.mark.parametrize("D,gemma", ADD_RMSNORM_GATHER_CASES)
def test_add_rmsnorm_gather(D, gemma):
    rows = 50
    h = randbf(rows, D, seed=24)
    w = randbf(D, seed=25)
    idx = torch.tensor([49, 0, 17], dtype=torch.int32, device=DEV)
    out = torch.empty(3, D, dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_add_rmsnorm(None, 0, 0, 0, None, p(h), D, p(w), p(out), D, p(idx), 3, D, 1e-6, gemma, st()) == 0
    sync()
    x = h[idx.long()].float()
    xhat = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    want = xhat * (1.0 + w.float()) if gemma else w.float() * rbf(xhat)
    assert_close_bf16(out, want, ulps=1.5, atol=1e-3, what="rmsnorm gather")


@pytest.mark.parametrize("permuted", [False, True])
def test_patchify_exact(permuted):
    nimg, H, W, patch, merge, tps = 2, 56, 84, 14, 2, 2
    kreal, kpad = 3 * tps * patch * patch, 1216
    g = torch.Generator().manual_seed(26)
    img = torch.randint(0, 256, (nimg, H, W, 3), generator=g, dtype=torch.uint8)
    lut = (torch.randn(3, 256, generator=g)).to(torch.bfloat16)
    gh, gw = H // patch, W // patch
    P = gh * gw
    ld = 64
    out = torch.full((nimg * ld, kpad), 7.0, dtype=torch.bfloat16, device=DEV)
    img_d, lut_d = img.to(DEV), lut.to(DEV)  # keep alive: the launch only sees raw pointers
    # permuted: output row r shows source patch row_src[r] (the window order of the Qwen2.5-VL tower)
    src = torch.randperm(P, generator=g).to(torch.int32) if permuted else torch.arange(P, dtype=torch.int32)
    src_d = src.to(DEV)
    rc = lib().hwocr_patchify(p(img_d), p(lut_d), p(out), nimg, H, W, patch, merge, tps, kpad, ld,
                              p(src_d) if permuted else None, st())
    assert rc == 0
    sync()
    # restatement of HF patchify on the LUT-mapped image (C,H,W)
    for im in range(nimg):
        chw = torch.stack([lut[c][img[im, :, :, c].long()] for c in range(3)])  # [3,H,W] bf16
        x = chw.reshape(3, gh // merge, merge, patch, gw // merge, merge, patch).permute(1, 4, 2, 5, 0, 3, 6)
        x = x.unsqueeze(5).expand(*x.shape[:5], tps, patch, patch).reshape(P, kreal)[src.long()]
        got = out[im * ld: im * ld + P].cpu()
        assert torch.equal(got[:, :kreal], x), "patchify must be an exact gather"
        assert (got[:, kreal:] == 0).all()


def _vit_pos(gh, gw, merge):
    hp = torch.arange(gh).view(-1, 1).expand(gh, gw)
    wp = torch.arange(gw).view(1, -1).expand(gh, gw)
    f = lambda t: t.reshape(gh // merge, merge, gw // merge, merge).permute(0, 2, 1, 3).reshape(-1)
    return f(hp), f(wp)


def _interleave_qk(qkv, heads, hd):
    """[rows][3][heads][hd] with the q / k features of every head as rotary pairs side by side (engine.interleave_rotary_pairs
    applied to the output features)."""
    rows = qkv.shape[0]
    v = qkv.view(rows, 3, heads, 2, hd // 2)
    il = v[:, :2].transpose(3, 4).reshape(rows, 2, heads, hd)
    return torch.cat([il, qkv.view(rows, 3, heads, hd)[:, 2:]], dim=1).reshape(rows, 3 * heads * hd).contiguous()


@pytest.mark.parametrize("interleaved", [0, 1])
@pytest.mark.parametrize("hd,heads", [(80, 4), (32, 2)])
def test_vit_rope_split(hd, heads, interleaved):
    gh, gw = 8, 12
    P = gh * gw
    tok_ld = 128
    D = heads * hd
    qkv = randbf(tok_ld, 3 * D, seed=27)
    qkv_in = _interleave_qk(qkv, heads, hd) if interleaved else qkv  # same values, pair-interleaved arrival order
    ph, pw = _vit_pos(gh, gw, 2)
    pos_h = torch.zeros(tok_ld, dtype=torch.int32)
    pos_w = torch.zeros(tok_ld, dtype=torch.int32)
    pos_h[:P], pos_w[:P] = ph, pw
    quarter = hd // 4
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
    tab = torch.arange(64, dtype=torch.float).unsqueeze(-1) * inv  # [64][quarter]
    cos_t, sin_t = tab.cos().contiguous(), tab.sin().contiguous()
    Q = torch.zeros(heads, tok_ld, hd, dtype=torch.bfloat16, device=DEV)
    K = torch.zeros_like(Q)
    VT = torch.full((heads, hd, tok_ld), float("nan"), dtype=torch.bfloat16, device=DEV)
    ph_d, pw_d, cos_d, sin_d = pos_h.to(DEV), pos_w.to(DEV), cos_t.to(DEV), sin_t.to(DEV)
    rc = lib().hwocr_vit_rope_split(p(qkv_in), p(Q), p(K), p(VT), p(ph_d), p(pw_d), p(cos_d), p(sin_d), P, tok_ld, heads,
                                    hd, interleaved, st())
    assert rc == 0
    sync()
    x = qkv[:P].float().cpu().view(P, 3, heads, hd)
    rot = torch.cat([tab[ph.long()], tab[pw.long()]], -1)  # [P, hd/2]
    emb = torch.cat([rot, rot], -1)
    cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)
    rh = lambda t: torch.cat([-t[..., hd // 2:], t[..., : hd // 2]], -1)
    for which, got in ((0, Q), (1, K)):
        want = x[:, which] * cos + rh(x[:, which]) * sin  # [P, heads, hd]
        assert_close_bf16(got[:, :P].cpu().permute(1, 0, 2), want, ulps=1.0, atol=1e-3, what=f"vit rope {which}")
    assert torch.equal(VT[:, :, :P].cpu(), qkv[:P].cpu().view(P, 3, heads, hd)[:, 2].permute(1, 2, 0))
    assert (VT[:, :, P:].cpu() == 0).all()
    assert quarter * 4 == hd


# (M, K, heads, hd): the page-read shape of the Qwen towers and SigLIP-padded (16 x 80), ragged M (edge tiles), other head sizes
VIT_QKV_CASES = [(5184, 1280, 16, 80), (1300, 256, 16, 80), (1088, 128, 8, 64), (2048, 192, 2, 128), (1024, 128, 16, 32),
                 (62208, 1280, 16, 80), (49152, 1152, 16, 80)]   # the last two: the bench's tower launches (Qwen2-VL / SigLIP-padded)


@pytest.mark.parametrize("fp8", [0, 1])
@pytest.mark.parametrize("M,K,heads,hd", VIT_QKV_CASES)
def test_gemm_vit_qkv_equals_gemm_then_rope_split(M, K, heads, hd, fp8):
    """hwocr_gemm_vit_qkv (rotary + head split + V transpose in the GEMM epilogue) against the two kernels it replaces,
    bit for bit, and against the fp32 restatement of HF's vision attention front end (modeling_qwen2_vl.py:239-248, :375-400)."""
    if fp8 and K % 128:
        pytest.skip("fp8 operands need K % 128 == 0")
    DH = heads * hd
    tok_ld = (M + 63) // 64 * 64
    x = randbf(M, K, seed=51)
    w = randbf(3 * DH, K, scale=K ** -0.5, seed=52)             # rows already in the pair-interleaved order
    bias = randbf(3 * DH, scale=0.5, seed=53)
    g = torch.Generator().manual_seed(3)
    pos_h = torch.randint(0, 60, (tok_ld,), generator=g, dtype=torch.int32)
    pos_w = torch.randint(0, 60, (tok_ld,), generator=g, dtype=torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
    tab = torch.arange(64, dtype=torch.float).unsqueeze(-1) * inv
    cos_d, sin_d, ph_d, pw_d = tab.cos().contiguous().to(DEV), tab.sin().contiguous().to(DEV), pos_h.to(DEV), pos_w.to(DEV)

    def bufs():
        return (torch.full((heads, tok_ld, hd), float("nan"), dtype=torch.bfloat16, device=DEV),
                torch.full((heads, tok_ld, hd), float("nan"), dtype=torch.bfloat16, device=DEV),
                torch.full((heads, hd, tok_ld), float("nan"), dtype=torch.bfloat16, device=DEV))

    from handwritten_ocr_amd import _lib
    if fp8:
        x8, xs = torch.empty(M, K, dtype=torch.uint8, device=DEV), torch.empty(M, dtype=torch.float32, device=DEV)
        w8, wsc = torch.empty(3 * DH, K, dtype=torch.uint8, device=DEV), torch.empty(3 * DH, dtype=torch.float32, device=DEV)
        assert lib().hwocr_quant_rows_fp8(p(x), p(x8), p(xs), M, K, K, K, st()) == 0
        assert lib().hwocr_quant_rows_fp8(p(w), p(w8), p(wsc), 3 * DH, K, K, K, st()) == 0
    # the two separate kernels
    qkv = torch.empty(M, 3 * DH, dtype=torch.bfloat16, device=DEV)
    if fp8:
        assert lib().hwocr_gemm_wide_fp8(p(x8), p(xs), p(w8), p(wsc), p(bias), None, p(qkv), M, 3 * DH, K, K, K, 3 * DH, 0, 0, st()) == 0
    else:
        assert lib().hwocr_gemm_wide(p(x), p(w), p(bias), None, p(qkv), M, 3 * DH, K, K, K, 3 * DH, 0, 0, st()) == 0
    Q0, K0, V0 = bufs()
    assert lib().hwocr_vit_rope_split(p(qkv), p(Q0), p(K0), p(V0), p(ph_d), p(pw_d), p(cos_d), p(sin_d), M, tok_ld, heads, hd, 1,
                                      st()) == 0
    # the fused one
    Q1, K1, V1 = bufs()
    sp = _lib.VitSplit(Q=p(Q1), K=p(K1), VT=p(V1), pos_h=p(ph_d), pos_w=p(pw_d), cos_tab=p(cos_d), sin_tab=p(sin_d), heads=heads,
                       hd=hd, tok_ld=tok_ld)
    import ctypes as C
    rc = lib().hwocr_gemm_vit_qkv(p(x8) if fp8 else p(x), p(w8) if fp8 else p(w), p(bias), M, K, K, K, p(xs) if fp8 else None,
                                  p(wsc) if fp8 else None, C.byref(sp), st())
    assert rc == 0
    sync()
    assert torch.equal(Q1[:, :M], Q0[:, :M]) and torch.equal(K1[:, :M], K0[:, :M]), "fused rotary differs from gemm + rope_split"
    assert torch.equal(V1[:, :, :M], V0[:, :, :M]), "fused V transpose differs from gemm + rope_split"
    if fp8:
        return  # the bf16 form below pins the arithmetic; the fp8 GEMM has its own exact test
    # fp32 restatement: de-interleave the features, rotate as HF does
    acc = rbf(x.float() @ w.float().t() + bias.float()).cpu().view(M, 3, heads, hd // 2, 2)
    qk = acc[:, :2].transpose(3, 4).reshape(M, 2, heads, hd)       # true feature order (d, then d + hd/2)
    rot = torch.cat([tab[pos_h[:M].long()], tab[pos_w[:M].long()]], -1)
    emb = torch.cat([rot, rot], -1)
    cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)
    rh = lambda t: torch.cat([-t[..., hd // 2:], t[..., : hd // 2]], -1)
    for which, got in ((0, Q1), (1, K1)):
        want = qk[:, which] * cos + rh(qk[:, which]) * sin
        # the bf16 rounding of acc + bias (a rounding point of the reference too) can flip by one ulp with the accumulation
        # order: 2 ulps of the larger of the pair
        mag = qk[:, which].abs() + rh(qk[:, which]).abs()
        assert_close_bf16(got[:, :M].cpu().permute(1, 0, 2), want, ulps=2.0, atol=2e-3, what=f"fused vit rope {which}", mag=mag)
    vwant = rbf(x.float() @ w.float().t() + bias.float()).cpu()[:, 2 * DH:].view(M, heads, hd).permute(1, 2, 0)
    assert_close_bf16(V1[:, :, :M].cpu(), vwant, ulps=1.0, atol=2e-3, what="fused V^T")


def test_gemm_vit_qkv_rejects_shapes_it_cannot_fuse():
    from handwritten_ocr_amd import _lib
    import ctypes as C
    x = randbf(64, 64)
    sp = _lib.VitSplit(Q=p(x), K=p(x), VT=p(x), pos_h=p(x), pos_w=p(x), cos_tab=p(x), sin_tab=p(x), heads=6, hd=64, tok_ld=1024)
    assert lib().hwocr_gemm_vit_qkv(p(x), p(x), None, 1024, 64, 64, 64, None, None, C.byref(sp), st()) == 1   # 6 x 64 is not whole tiles
    sp.heads = 8
    assert lib().hwocr_gemm_vit_qkv(p(x), p(x), None, 512, 64, 64, 64, None, None, C.byref(sp), st()) == 1    # too few rows


def _mrope_ref(x, pos3, cos_tab, sin_tab, sec0, sec1):
    # x [rows, heads, hd] fp32 (bf16 values); tables bf16 [maxpos][hd/2]; returns fp32 of the bf16 result
    half = x.shape[-1] // 2
    i = torch.arange(half)
    axis = torch.where(i < sec0, 0, torch.where(i < sec1, 1, 2))
    pidx = pos3[axis, :].t().long()  # [rows, half]
    cs = cos_tab.float()[pidx, i].unsqueeze(1)  # [rows,1,half]
    sn = sin_tab.float()[pidx, i].unsqueeze(1)
    x1, x2 = x[..., :half], x[..., half:]
    oa = rbf(rbf(x1 * cs) + rbf(-x2 * sn))
    ob = rbf(rbf(x2 * cs) + rbf(x1 * sn))
    return torch.cat([oa, ob], -1)


def _rope_tables(maxpos, theta=1e6, hd=128):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    fr = torch.arange(maxpos, dtype=torch.float).unsqueeze(-1) * inv
    return fr.cos().to(torch.bfloat16), fr.sin().to(torch.bfloat16)


# hd 256 with sec0 = 128: plain RoPE on the first position axis (Gemma)
@pytest.mark.parametrize("hd,sec0,sec1,tiled", MROPE_CASES)
def test_mrope_kv_prefill(hd, sec0, sec1, tiled):
    Hq, Hkv, nseq, Tp, ctx = 4, 2, 2, 128, 256
    rows = nseq * Tp
    W = (Hq + 2 * Hkv) * hd
    qkv = randbf(rows, W, seed=28)
    g = torch.Generator().manual_seed(29)
    pos3 = torch.randint(0, 300, (3, rows), generator=g, dtype=torch.int32)
    cos_t, sin_t = _rope_tables(512, hd=hd)
    Q = torch.zeros(rows, Hq * hd, dtype=torch.bfloat16, device=DEV)
    Kc = torch.zeros(nseq, Hkv, ctx, hd, dtype=torch.bfloat16, device=DEV)
    VT = torch.zeros(nseq, Hkv, hd, ctx, dtype=torch.bfloat16, device=DEV)
    pos_d, cos_d, sin_d = pos3.to(DEV), cos_t.to(DEV), sin_t.to(DEV)
    rc = lib().hwocr_mrope_kv_prefill(p(qkv), p(Q), p(Kc), p(VT), p(pos_d), p(cos_d), p(sin_d),
                                      rows, Tp, Hq, Hkv, sec0, sec1, Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx,
                                      ctx, hd, tiled, st())
    assert rc == 0
    sync()
    if tiled:
        Kc, VT = untile_k(Kc), untile_v(VT)
    x = qkv.float().cpu()
    qw = _mrope_ref(x[:, : Hq * hd].view(rows, Hq, hd), pos3, cos_t, sin_t, sec0, sec1)
    kw = _mrope_ref(x[:, Hq * hd: (Hq + Hkv) * hd].view(rows, Hkv, hd), pos3, cos_t, sin_t, sec0, sec1)
    assert torch.equal(Q.float().cpu().view(rows, Hq, hd), qw), "M-RoPE q must reproduce the bf16 rounding chain"
    assert torch.equal(Kc[:, :, :Tp].float().cpu(), kw.view(nseq, Tp, Hkv, hd).permute(0, 2, 1, 3))
    vw = qkv.cpu()[:, (Hq + Hkv) * hd:].view(nseq, Tp, Hkv, hd).permute(0, 2, 3, 1)
    assert torch.equal(VT[:, :, :, :Tp].cpu(), vw)


@pytest.mark.parametrize("hd,Hq,Hkv,sec0,sec1,tiled", [(128, 12, 2, 16, 40, 0), (128, 12, 2, 16, 40, 1), (256, 8, 1, 128, 128, 0)])
def test_decode_qkv_finish(hd, Hq, Hkv, sec0, sec1, tiled):
    B, ctx, nslab = 5, 256, 3
    W = (Hq + 2 * Hkv) * hd
    slabs = randf32(nslab, B, W, seed=903)
    bias = randbf(W, seed=30)
    lens = torch.tensor([1, 10, 200, 256, 77], dtype=torch.int32)
    delta = torch.tensor([0, -5, 3, -100, 40], dtype=torch.int32)
    cos_t, sin_t = _rope_tables(512, hd=hd)
    Q = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    Kc = torch.zeros(B, Hkv, ctx, hd, dtype=torch.bfloat16, device=DEV)
    VT = torch.zeros(B, Hkv, hd, ctx, dtype=torch.bfloat16, device=DEV)
    lens_d, delta_d, cos_d, sin_d = lens.to(DEV), delta.to(DEV), cos_t.to(DEV), sin_t.to(DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = lib().hwocr_decode_qkv_finish(p(slabs), nslab, B * W, p(bias), p(Q), p(Kc), p(VT), p(lens_d),
                                       p(delta_d), p(cos_d), p(sin_d), B, Hq, Hkv,
                                       Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx, hd, tiled, ctx, 512,
                                       p(status), st())
    assert rc == 0
    sync()
    assert int(status) == 0
    if tiled:
        Kc, VT = untile_k(Kc), untile_v(VT)
    row = rbf((slabs.sum(0) + bias.float()).cpu())
    pos = (lens - 1 + delta)
    pos3 = pos.unsqueeze(0).expand(3, B)
    qw = _mrope_ref(row[:, : Hq * hd].view(B, Hq, hd), pos3, cos_t, sin_t, sec0, sec1)
    kw = _mrope_ref(row[:, Hq * hd: (Hq + Hkv) * hd].view(B, Hkv, hd), pos3, cos_t, sin_t, sec0, sec1)
    # slab summation order differs from torch.sum by fp32 rounding only -> 1 bf16 ulp
    assert_close_bf16(Q.view(B, Hq, hd).cpu(), qw, ulps=1.0, atol=1e-3, what="decode q")
    for b in range(B):
        slot = int(lens[b]) - 1
        assert_close_bf16(Kc[b, :, slot].cpu(), kw[b], ulps=1.0, atol=1e-3, what="decode k")
        assert_close_bf16(VT[b, :, :, slot].cpu(), row[b, (Hq + Hkv) * hd:].view(Hkv, hd), ulps=1.0, atol=1e-3,
                          what="decode v")
        assert (Kc[b].float().abs().sum(-1) != 0).sum() == Hkv  # exactly one slot written per kv head


@pytest.mark.parametrize("tiled", [0, 1])
def test_decode_qkv_finish_flags_reads_outside_their_invariants(tiled):
    """A read whose position lens - 1 + rope_delta falls before / past the rope table, or whose cache slot is outside the
    cache (a parked slot that kept the negative rope_delta of the read it held; a length that ran past ctx), is skipped —
    nothing of it is written — and HWOCR_STATUS_BAD_POSITION is raised; the healthy reads of the same launch are served.
    Round 1 clamped the position to 0 instead, which hid exactly this host bug (commit 7665a5c)."""
    hd, Hq, Hkv, B, ctx, nslab, max_pos = 128, 4, 2, 6, 128, 2, 160
    W = (Hq + 2 * Hkv) * hd
    slabs = randf32(nslab, B, W, seed=904)
    #          ok   parked, stale -1295   ok    slot past ctx   position past the table   lens 0
    lens = [10, 1, 128, 129, 100, 0]
    delta = [-3, -1295, 31, 0, 80, 0]
    bad = [False, True, False, True, True, True]
    cos_t, sin_t = _rope_tables(max_pos, hd=hd)
    Q = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    Kc = torch.zeros(B, Hkv, ctx, hd, dtype=torch.bfloat16, device=DEV)
    VT = torch.zeros(B, Hkv, hd, ctx, dtype=torch.bfloat16, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    delta_d = torch.tensor(delta, dtype=torch.int32, device=DEV)
    cos_d, sin_d = cos_t.to(DEV), sin_t.to(DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = lib().hwocr_decode_qkv_finish(p(slabs), nslab, B * W, None, p(Q), p(Kc), p(VT), p(lens_d), p(delta_d), p(cos_d),
                                       p(sin_d), B, Hq, Hkv, Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx, hd, tiled,
                                       ctx, max_pos, p(status), st())
    assert rc == 0
    sync()
    assert int(status) == 1
    for b in range(B):
        touched = bool(Q[b].float().abs().sum() != 0) or bool(Kc[b].float().abs().sum() != 0) or bool(VT[b].float().abs().sum() != 0)
        assert touched != bad[b], f"read {b}: lens {lens[b]} delta {delta[b]}"
    # without a status word the bad reads are still skipped
    Q.zero_()
    assert lib().hwocr_decode_qkv_finish(p(slabs), nslab, B * W, None, p(Q), p(Kc), p(VT), p(lens_d), p(delta_d), p(cos_d),
                                         p(sin_d), B, Hq, Hkv, Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx, hd,
                                         tiled, ctx, max_pos, None, st()) == 0
    sync()
    assert all(bool(Q[b].float().abs().sum() != 0) != bad[b] for b in range(B))


@pytest.mark.parametrize("hd,Hq,Hkv,tiled,nsplit,B", [(128, 12, 2, 1, 1, 9), (128, 12, 2, 1, 4, 5), (128, 28, 4, 0, 1, 6),
                                                      (128, 6, 2, 0, 3, 4), (256, 8, 1, 0, 1, 5), (256, 8, 1, 0, 4, 5),
                                                      (128, 12, 2, 1, 1, 252)])
def test_attn_decode_qkv_equals_finish_then_attention(hd, Hq, Hkv, tiled, nsplit, B):
    """hwocr_attn_decode_qkv (slab sum + bias + rotary + cache append inside the attention launch) against the two launches
    it replaces: same attention output, same cache, bit for bit — including reads whose new slot opens a fresh 32-key block and
    reads at the first and the last cache position."""
    ctx, nslab, max_pos = 512, 3, 1024
    W = (Hq + 2 * Hkv) * hd
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(2, ctx, (B,), generator=g).tolist()
    lens[0], lens[1], lens[2], lens[3] = 1, ctx, 33, 64            # first slot, last slot, first of a block, last of a block
    delta = torch.randint(-1, 300, (B,), generator=g).tolist()
    delta[0] = 0
    slabs = randf32(nslab, B, W, seed=905)
    bias = randbf(W, seed=30) if hd == 128 else None               # Gemma projections carry no bias
    cos_t, sin_t = _rope_tables(max_pos, hd=hd)
    k = randbf(B, Hkv, ctx, hd, seed=16)
    v = randbf(B, Hkv, ctx, hd, seed=17)
    vt = v.transpose(2, 3).contiguous()
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    delta_d = torch.tensor(delta, dtype=torch.int32, device=DEV)
    cos_d, sin_d = cos_t.to(DEV), sin_t.to(DEV)
    G = Hq // Hkv
    strides = (Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx)

    def caches():
        return (tile_k(k), tile_v(vt)) if tiled else (k.clone(), vt.clone())

    def parts():
        return (torch.zeros(B * Hkv * nsplit * G * hd, dtype=torch.float32, device=DEV),
                torch.zeros(B * Hkv * nsplit * G * 2, dtype=torch.float32, device=DEV))

    # two launches
    K0, V0 = caches()
    Q0 = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    st0 = torch.zeros(1, dtype=torch.int32, device=DEV)
    po, pm = parts()
    out0 = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_decode_qkv_finish(p(slabs), nslab, B * W, p(bias), p(Q0), p(K0), p(V0), p(lens_d), p(delta_d), p(cos_d),
                                         p(sin_d), B, Hq, Hkv, *strides, hd, tiled, ctx, max_pos, p(st0), st()) == 0
    assert lib().hwocr_attn_decode(p(Q0), p(K0), p(V0), p(lens_d), p(out0), p(po), p(pm), None, B, Hq, Hkv, nsplit, *strides,
                                   hd ** -0.5, hd, tiled, st()) == 0
    # one launch; with splits, once with the merge launch and once with the last workgroup merging (arrival counters)
    for lastwg in ([False, True] if nsplit > 1 else [False]):
        K1, V1 = caches()
        st1 = torch.zeros(1, dtype=torch.int32, device=DEV)
        po1, pm1 = parts()
        arrive = torch.zeros(B * Hkv, dtype=torch.int32, device=DEV) if lastwg else None
        out1 = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_attn_decode_qkv(p(slabs), nslab, B * W, p(bias), p(K1), p(V1), p(lens_d), p(delta_d), p(cos_d), p(sin_d),
                                           p(out1), p(po1), p(pm1), p(arrive), B, Hq, Hkv, nsplit, *strides, hd ** -0.5, hd, tiled,
                                           ctx, max_pos, p(st1), st()) == 0
        sync()
        assert int(st0) == 0 and int(st1) == 0
        assert torch.equal(K1, K0) and torch.equal(V1, V0), "cache after the fused launch differs"
        assert torch.equal(out1, out0), "attention output of the fused launch differs"
        assert bool(torch.isfinite(out1.float()).all())
        assert arrive is None or int(arrive.abs().sum()) == 0


def test_attn_decode_qkv_flags_reads_outside_their_invariants():
    hd, Hq, Hkv, B, ctx, nslab, max_pos = 128, 4, 2, 4, 128, 2, 160
    W = (Hq + 2 * Hkv) * hd
    slabs = randf32(nslab, B, W, seed=906)
    lens = torch.tensor([10, 1, 129, 100], dtype=torch.int32, device=DEV)       # ok, stale negative delta, slot past ctx, pos past table
    delta = torch.tensor([-3, -1295, 0, 80], dtype=torch.int32, device=DEV)
    cos_t, sin_t = _rope_tables(max_pos, hd=hd)
    Kc = torch.zeros(B, Hkv, ctx, hd, dtype=torch.bfloat16, device=DEV)
    VT = torch.zeros(B, Hkv, hd, ctx, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(B, Hq * hd, dtype=torch.bfloat16, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    cos_d, sin_d = cos_t.to(DEV), sin_t.to(DEV)
    assert lib().hwocr_attn_decode_qkv(p(slabs), nslab, B * W, None, p(Kc), p(VT), p(lens), p(delta), p(cos_d), p(sin_d), p(out),
                                       None, None, None, B, Hq, Hkv, 1, Hkv * ctx * hd, ctx * hd, Hkv * hd * ctx, hd * ctx, ctx,
                                       hd ** -0.5, hd, 0, ctx, max_pos, p(status), st()) == 0
    sync()
    assert int(status) == 1
    touched = [bool(Kc[b].float().abs().sum() != 0) or bool(out[b].float().abs().sum() != 0) for b in range(B)]
    assert touched == [True, False, False, False]


def test_embed_splice():
    V, D, rows = 100, 256, 20
    table = randbf(V, D, seed=31)
    img = randbf(6, D, seed=32)
    ids = torch.arange(rows, dtype=torch.int32) % V
    img_row = torch.full((rows,), -1, dtype=torch.int32)
    img_row[3:9] = torch.arange(6, dtype=torch.int32)
    out = torch.zeros(rows, D, dtype=torch.bfloat16, device=DEV)
    ids_d, row_d = ids.to(DEV), img_row.to(DEV)
    assert lib().hwocr_embed_splice(p(ids_d), p(row_d), p(table), p(img), p(out), rows, D, 1.0, st()) == 0
    sync()
    want = table[ids.long().to(DEV)].clone()
    want[3:9] = img
    assert torch.equal(out, want)


def _select_ws(split, B):
    """hwocr_argmax_advance's split_ws (hwocr.h): zero counters -> a row is scanned by 16 workgroups and finished by the last one."""
    from handwritten_ocr_amd import _lib
    return torch.zeros(B * _lib.SELECT_WS_INTS, dtype=torch.int32, device=DEV) if split else None


# split: the 16-workgroups-per-read form (<= 16 reads and a row of >= 32768 ids), same cases at the model's vocabulary size
@pytest.mark.parametrize("V,split", [(1000, False), (151936, False), (151936, True)])
def test_argmax_advance_semantics(V, split):
    import ctypes as C
    B, max_new, E = 4, 8, V - 1
    logits = randbf(B, V, seed=33)
    logits[0, 10] = 50.0
    logits[0, V - 20] = 50.0   # tie (in the split form: across workgroups) -> lowest index (torch.argmax picks the first maximum)
    logits[1, E] = 60.0    # EOS wins but n_gen < min_new -> suppressed
    logits[1, 5] = 55.0
    logits[2, E] = 60.0    # EOS, allowed
    logits[3, 7] = 60.0    # already finished -> pad
    cur = torch.zeros(B, dtype=torch.int32, device=DEV)
    lens = torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=DEV)
    n_gen = torch.tensor([0, 0, 3, 3], dtype=torch.int32, device=DEV)
    fin = torch.tensor([0, 0, 0, 1], dtype=torch.int32, device=DEV)
    outt = torch.full((B, max_new), -1, dtype=torch.int32, device=DEV)
    eos = (C.c_int * 4)(E, 0, 0, 0)
    ws = _select_ws(split, B)
    rc = lib().hwocr_argmax_advance(p(logits), V, V, B, p(cur), p(lens), p(n_gen), p(fin), p(outt), max_new, 2, eos, 1,
                                    123, None, 0, 1.0, p(ws), st())
    assert rc == 0
    sync()
    assert cur.tolist() == [10, 5, E, 123]
    assert lens.tolist() == [6, 7, 8, 9]
    assert n_gen.tolist() == [1, 1, 4, 4]
    assert fin.tolist() == [0, 0, 1, 1]
    assert outt[0, 0] == 10 and outt[1, 0] == 5 and outt[2, 3] == E and outt[3, 3] == 123
    assert ws is None or int(ws.view(B, -1)[:, 0].abs().sum()) == 0   # the arrival counters are zero again


@pytest.mark.parametrize("V,split", [(1024, False), (151936, False), (151936, True)])
def test_argmax_repetition_penalty(V, split):
    """HF RepetitionPenaltyLogitsProcessor inside the select kernel: ids in the bitmap have their fp32 score divided
    (positive) or multiplied (negative) by the penalty; the token a step was fed joins the bitmap first (not after a
    prefill, n_gen == 0; not for a finished read)."""
    import ctypes as C
    B, max_new, pen, far = 4, 4, 1.5, V - 124
    logits = torch.full((B, V), -3.0, dtype=torch.bfloat16, device=DEV)
    logits[0, 40], logits[0, 41] = 6.0, 5.0      # 40 is in the bitmap: 6 / 1.5 = 4 < 5 -> 41 wins
    logits[1, :] = -8.0
    logits[1, 7], logits[1, far] = -2.0, -2.5    # 7 is the token this step was fed: -2 * 1.5 = -3 < -2.5 -> `far` wins
    logits[2, 77] = 9.0                          # finished read: pad, bitmap untouched
    logits[3, 12], logits[3, 13] = 6.0, 5.0      # n_gen == 0 (first pick after a prefill): the stale cur_id 12 is NOT added
    seen = torch.zeros(B, V // 32, dtype=torch.int32, device=DEV)
    seen[0, 40 // 32] = 1 << (40 % 32)
    cur = torch.tensor([3, 7, 77, 12], dtype=torch.int32, device=DEV)
    lens = torch.tensor([3, 3, 3, 3], dtype=torch.int32, device=DEV)
    n_gen = torch.tensor([2, 1, 1, 0], dtype=torch.int32, device=DEV)
    fin = torch.tensor([0, 0, 1, 0], dtype=torch.int32, device=DEV)
    outt = torch.full((B, max_new), -1, dtype=torch.int32, device=DEV)
    eos = (C.c_int * 4)(V - 1, 0, 0, 0)
    ws = _select_ws(split, B)
    assert lib().hwocr_argmax_advance(p(logits), V, V, B, p(cur), p(lens), p(n_gen), p(fin), p(outt), max_new, 0, eos, 1, 5,
                                      p(seen), V // 32, pen, p(ws), st()) == 0
    sync()
    assert cur.tolist() == [41, far, 5, 12]
    s = seen.cpu()
    assert int(s[0, 0]) == 1 << 3 and int(s[0, 1]) == 1 << 8      # fed token 3 added; the new pick 41 is not (yet)
    assert int(s[1, 0]) == 1 << 7 and int(s[1].abs().sum()) == 1 << 7
    assert int(s[2].abs().sum()) == 0 and int(s[3].abs().sum()) == 0
    # penalty 1: nothing is rescaled
    cur.copy_(torch.tensor([3, 7, 77, 12], dtype=torch.int32))
    assert lib().hwocr_argmax_advance(p(logits), V, V, B, p(cur), p(lens), p(n_gen), p(fin), p(outt), max_new, 0, eos, 1, 5,
                                      p(seen), V // 32, 1.0, p(ws), st()) == 0
    sync()
    assert cur.tolist()[:2] == [40, 7]
    assert ws is None or int(ws.view(B, -1)[:, 0].abs().sum()) == 0


@pytest.mark.parametrize("B", [1, 3, 16])
def test_argmax_split_form_picks_what_the_one_workgroup_form_picks(B):
    """Random rows at the model's vocabulary size with planted ties, a penalty bitmap, EOS suppression: 16 workgroups per read + the
    last one finishing (split_ws) against one workgroup per read, over three consecutive steps (the fed token joins the bitmap)."""
    import ctypes as C
    V, max_new = 151936, 4
    g = torch.Generator().manual_seed(7 + B)
    eos = (C.c_int * 4)(V - 1, 17, 0, 0)

    def run(split):
        seen = torch.zeros(B, V // 32, dtype=torch.int32, device=DEV)
        seen[:, ::7] = 0x10101
        cur = torch.zeros(B, dtype=torch.int32, device=DEV)
        lens = torch.full((B,), 9, dtype=torch.int32, device=DEV)
        n_gen = torch.zeros(B, dtype=torch.int32, device=DEV)
        fin = torch.zeros(B, dtype=torch.int32, device=DEV)
        outt = torch.full((B, max_new), -1, dtype=torch.int32, device=DEV)
        ws = _select_ws(split, B)
        gg = torch.Generator().manual_seed(11 + B)
        for step in range(3):
            logits = (torch.randn(B, V, generator=gg) * 3).to(torch.bfloat16)
            top = logits.float().max(dim=1).values
            for b in range(B):  # the maximum again far away (a tie across workgroups) and on an EOS id (suppressed while n_gen < 2)
                logits[b, (b * 9973 + 140000) % V] = top[b]
                logits[b, 17] = top[b] + 1
            logits = logits.to(DEV)
            assert lib().hwocr_argmax_advance(p(logits), V, V, B, p(cur), p(lens), p(n_gen), p(fin), p(outt), max_new, 2, eos, 2, 0,
                                              p(seen), V // 32, 1.3, p(ws), st()) == 0
            sync()
        return cur.cpu(), outt.cpu(), fin.cpu(), seen.cpu(), n_gen.cpu()

    want, got = run(False), run(True)
    for w, g_ in zip(want, got):
        assert torch.equal(w, g_)
    assert int(want[2].sum()) == B   # step 3 (n_gen == 2 == min_new) picks the EOS id 17: every read finished


# ---------------------------------------------------------------------------------------------- fp8 (E4M3) wide GEMM
def _quant_gpu(x):
    rows, K = x.shape
    q = torch.full((rows, K), 0x7F, dtype=torch.uint8, device=DEV)  # NaN pattern: every byte must be overwritten
    s = torch.full((rows,), float("nan"), dtype=torch.float32, device=DEV)
    assert lib().hwocr_quant_rows_fp8(p(x), p(q), p(s), rows, K, K, K, st()) == 0
    sync()
    return q, s


@pytest.mark.parametrize("rows,K", [(1, 128), (37, 1152), (260, 4352), (5, 16384)])
def test_quant_rows_fp8_bit_exact(rows, K):
    from oracle import fp8_ref
    x = randbf(rows, K, scale=2.0, seed=11)
    x[0, : K // 2] *= 40.0                      # wide dynamic range inside one row: subnormal codes appear
    if rows > 2:
        x[2] = 0                                # all-zero row: scale 1, codes 0
    q, s = _quant_gpu(x)
    wq, ws = fp8_ref.quant_rows(x.cpu())
    assert torch.equal(s.cpu(), ws), "row scales differ"
    assert torch.equal(q.cpu(), wq.view(torch.uint8)), "E4M3 codes differ"


WIDE_FP8_SHAPES = [(64, 128, 128), (300, 264, 256), (1000, 384, 1152), (2065, 1024, 640), (4096, 1152, 4352)]
WIDE_FP8_EPIS = [0, 1, 2, 6]
# the fp8 bench's launches (PaliGemma-3B, 12 pages / 16 prompts per launch) whose workgroups walk several tiles, + a ragged one
WIDE_FP8_BENCH_CASES = [(49152, 4352, 1152, 6), (49152, 1152, 4352, 1), (49152, 1152, 1280, 1), (66560, 2560, 2048, 0), (66560, 2048, 2048, 1),
                        (12288, 4352, 1152, 6), (20000, 2048, 16384, 1),
                        (15552, 5120, 1280, 2)]   # quick-GELU in E4M3: `bench.py --fp8` on the Qwen2-VL tower
WIDE_FP8_GATED_SHAPES = [(1328, 1792, 1536), (8192, 8192, 2048)]  # the second: 1024 tiles


def _fp8_gemm_ref(xq, xs, wq, ws):
    """oracle/fp8_ref.gemm's arithmetic on the device (the float64 product of large shapes takes minutes on the host)."""
    xd = xq.view(torch.float8_e4m3fn).float().double()
    wd = wq.view(torch.float8_e4m3fn).float().double()
    return (xd @ wd.t()).float() * (ws[None, :] * xs[:, None])


def test_fp8_gemm_ref_on_the_device_equals_the_oracle():
    from oracle import fp8_ref
    xq, xs = _quant_gpu(randbf(70, 256, seed=3))
    wq, ws = _quant_gpu(randbf(40, 256, scale=0.1, seed=4))
    want = fp8_ref.gemm(xq.cpu().view(torch.float8_e4m3fn), xs.cpu(), wq.cpu().view(torch.float8_e4m3fn), ws.cpu())
    assert torch.equal(_fp8_gemm_ref(xq, xs, wq, ws).cpu(), want)


@pytest.mark.parametrize("M,N,K,epi", [(m, n, k, e) for (m, n, k) in WIDE_FP8_SHAPES for e in WIDE_FP8_EPIS] + WIDE_FP8_BENCH_CASES)
def test_gemm_wide_fp8(M, N, K, epi):
    """The kernel against the exact scaled product of the SAME codes (oracle/fp8_ref.py): only fp32 accumulation order and
    the bf16 epilogue rounding separate them."""
    x = randbf(M, K, scale=1.0, seed=21)
    w = randbf(N, K, scale=K ** -0.5, seed=22)
    bias = randbf(N, scale=0.5, seed=23)
    res = randbf(M, N, seed=24)
    xq, xs = _quant_gpu(x)
    wq, ws = _quant_gpu(w)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    rc = lib().hwocr_gemm_wide_fp8(p(xq), p(xs), p(wq), p(ws), p(bias), p(res) if epi == 1 else None, p(out), M, N, K, K, K, N,
                                   N, epi, st())
    assert rc == 0
    sync()
    acc = _fp8_gemm_ref(xq, xs, wq, ws)
    _check_epilogue(out, acc, bias, res, epi, f"gemm_wide_fp8 epi={epi} {M}x{N}x{K}")   # (2 ulps; explained outliers only)
    # and the quantisation itself stays where E4M3 puts it: a few percent of the bf16 product's spread
    exact = x.float() @ w.float().t()
    assert float((acc - exact).abs().mean() / exact.abs().mean()) < 0.06


@pytest.mark.parametrize("M,N,K", WIDE_FP8_GATED_SHAPES)
@pytest.mark.parametrize("geglu", [False, True])
def test_gemm_wide_fp8_gated(geglu, M, N, K):
    x = randbf(M, K, seed=25)
    w = randbf(N, K, scale=K ** -0.5, seed=26)
    xq, xs = _quant_gpu(x)
    wq, ws = _quant_gpu(w)
    out = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_gemm_wide_fp8(p(xq), p(xs), p(wq), p(ws), None, None, p(out), M, N, K, K, K, N // 2, 0, 7 if geglu else 4,
                                     st()) == 0
    sync()
    _check_gated(out, _fp8_gemm_ref(xq, xs, wq, ws), None, geglu, f"gemm_wide_fp8 gated {M}x{N}x{K}")


# ---------------------------------------------------------------------------------------------- E4M3 decode GEMMs (weight-only fp8)
def _tiled_fp8(wq):
    n, k = wq.shape
    out = torch.full((n * k,), 0x7F, dtype=torch.uint8, device=DEV)
    assert lib().hwocr_tile_weights_fp8(p(wq), p(out), n, k, k, st()) == 0
    sync()
    return out


def test_tile_weights_fp8_layout():
    n, k = 48, 192
    w = (torch.arange(n * k, dtype=torch.int64) % 251).to(torch.uint8).view(n, k).to(DEV)
    t = _tiled_fp8(w).view(n // 16, k // 64, 4, 16, 2, 8).cpu()          # [tile][k tile][q][c][half][8]
    want = w.cpu().view(n // 16, 16, k // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5)   # [tile][kt][q][c][half][8]
    assert torch.equal(t, want)


# the decode GEMMs of the fp8 configuration exactly as hwocr_decode_step issues them: PaliGemma-3B (config 4), the Qwen2-VL-2B
# shape under --fp8, the tiny golden model; (N, K, epi, splitk) from engine.decode_plan(..., fp8=True)
DECODE_GEMM_SHAPES_W8 = {
    3: [(2560, 2048, 5, 5), (32768, 2048, 7, 1), (2048, 16384, 5, 12), (257216, 2048, 0, 1), (17920, 1536, 4, 1), (512, 256, 0, 1)],
    24: [(2560, 2048, 5, 5), (32768, 2048, 7, 1), (1024, 256, 7, 1), (17920, 1536, 4, 1), (257216, 2048, 0, 1)],
    40: [(2048, 2048, 5, 5), (32768, 2048, 7, 1)],
    126: [(2560, 2048, 5, 5), (2048, 2048, 5, 5), (32768, 2048, 7, 1), (2048, 16384, 5, 12), (257216, 2048, 0, 1), (17920, 1536, 4, 1)],
    252: [(2560, 2048, 5, 4), (2048, 2048, 5, 4), (32768, 2048, 7, 1), (2048, 16384, 5, 8), (257216, 2048, 0, 1),
          (2048, 1536, 5, 4), (17920, 1536, 4, 1), (1536, 8960, 5, 8), (151936, 1536, 0, 1), (37888, 3584, 4, 1),
          (512, 256, 0, 1), (1024, 256, 7, 1), (256, 512, 5, 1), (512, 256, 4, 1)],
}
DECODE_GEMM_CASES_W8 = [(B,) + shape for B, shapes in DECODE_GEMM_SHAPES_W8.items() for shape in shapes]


@pytest.mark.parametrize("B,N,K,epi,splitk", DECODE_GEMM_CASES_W8)
def test_gemm_skinny_w8_decode_shapes(B, N, K, epi, splitk):
    """hwocr_gemm_skinny_w8 = bf16 activations x E4M3 weight codes (one fp32 scale per output feature) against the exact product
    of the SAME codes (oracle/fp8_ref.py decodes them): only the fp32 accumulation order and the bf16 epilogue separate the two."""
    x = randbf(B, K, seed=80)
    w = randbf(N, K, scale=K ** -0.5, seed=81)
    wq, ws = _quant_gpu(w)
    wt = _tiled_fp8(wq)
    wdq = wq.view(torch.float8_e4m3fn).float()                           # the values the codes stand for
    acc = (x.float() @ wdq.t()) * ws[None, :]
    if epi == 5:
        slabs = torch.full((splitk, B, N), float("nan"), dtype=torch.float32, device=DEV)
        assert lib().hwocr_gemm_skinny_w8(p(x), p(wt), p(ws), None, p(slabs), B, N, K, K, N, 5, splitk, st()) == 0
        sync()
        got = slabs.sum(0)
        assert torch.isfinite(got).all(), "a slab element was left unwritten"
        assert torch.allclose(got, acc, rtol=1e-4, atol=2e-3), f"w8 partial splitk={splitk}: {(got - acc).abs().max()}"
    elif epi == 0:
        bias = randbf(N, scale=0.5, seed=82)
        out = torch.full((B, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_skinny_w8(p(x), p(wt), p(ws), p(bias), p(out), B, N, K, K, N, 0, 1, st()) == 0
        sync()
        assert_close_bf16(out, acc + bias.float(), ulps=2.0, atol=2e-3, what="w8 linear")
    else:
        out = torch.full((B, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert lib().hwocr_gemm_skinny_w8(p(x), p(wt), p(ws), None, p(out), B, N, K, K, N // 2, epi, 1, st()) == 0
        sync()
        want = _swiglu_ref(acc, geglu=(epi == 7))
        gate = rbf(acc.view(B, N // 32, 2, 16)[:, :, 0, :]).reshape(B, N // 2)
        assert_close_bf16(out, want, ulps=3.0, atol=2e-3, what=f"w8 glu epi={epi}", mag=want.abs() * (1.0 + gate.abs()))
    # weight-only E4M3 stays within a few percent of the bf16 product's spread
    exact = x.float() @ w.float().t()
    assert float((acc - exact).abs().mean() / exact.abs().mean()) < 0.05


def test_gemm_skinny_w8_rejects_bad_shapes():
    z = torch.zeros(64, 64, dtype=torch.uint8, device=DEV)
    f = torch.zeros(64, dtype=torch.float32, device=DEV)
    x = randbf(4, 96)
    assert lib().hwocr_gemm_skinny_w8(p(x), p(z), p(f), None, p(x), 4, 64, 96, 96, 64, 0, 1, st()) == 1     # K % 64
    assert lib().hwocr_gemm_skinny_w8(p(x), p(z), None, None, p(x), 4, 64, 64, 64, 64, 0, 1, st()) == 1      # no scales
    assert lib().hwocr_tile_weights_fp8(p(z), p(z), 40, 64, 64, st()) == 1                                  # N % 16


def test_gemm_wide_fp8_rejects_bad_shapes():
    z = torch.zeros(256, 256, dtype=torch.uint8, device=DEV)
    s = torch.ones(256, dtype=torch.float32, device=DEV)
    o = torch.zeros(256, 256, dtype=torch.bfloat16, device=DEV)
    assert lib().hwocr_gemm_wide_fp8(p(z), p(s), p(z), p(s), None, None, p(o), 256, 256, 192, 256, 256, 256, 0, 0, st()) == 1
    assert lib().hwocr_gemm_wide_fp8(p(z), None, p(z), p(s), None, None, p(o), 256, 256, 256, 256, 256, 256, 0, 0, st()) == 1
    assert lib().hwocr_quant_rows_fp8(p(o), p(z), p(s), 256, 100, 256, 256, st()) == 1


# > 512 rows: hwocr_add_rmsnorm then runs the same wave-per-row kernel as the fp8 form (the few-row kernel sums in another order)
# (49152, 1152): the SigLIP tower launch of the fp8 bench (three trips of the LayerNorm's grid-stride loop); (66560, 2048): its prefill
NORM_FP8_CASES = [(600, 128), (1000, 1152), (700, 2048), (520, 3584), (49152, 1152), (66560, 2048), (700, 768)]


@pytest.mark.parametrize("rows,D", NORM_FP8_CASES)
def test_norms_emitting_fp8_equal_norm_then_quantise(rows, D):
    """hwocr_layernorm_fp8 / hwocr_rmsnorm_fp8 = the bf16 norm followed by hwocr_quant_rows_fp8, bit for bit."""
    x = randbf(rows, D, scale=1.5, seed=31)
    w = randbf(D, scale=0.3, seed=32) + 1.0
    b = randbf(D, scale=0.2, seed=33)
    xn = torch.empty(rows, D, dtype=torch.bfloat16, device=DEV)
    for kind in ("layernorm", "rms", "rms_gemma"):
        q = torch.full((rows, D), 0x7F, dtype=torch.uint8, device=DEV)
        s = torch.full((rows,), float("nan"), dtype=torch.float32, device=DEV)
        if kind == "layernorm":
            assert lib().hwocr_layernorm(p(x), p(w), p(b), p(xn), rows, D, D, D, 1e-6, st()) == 0
            assert lib().hwocr_layernorm_fp8(p(x), p(w), p(b), p(q), p(s), rows, D, D, D, 1e-6, st()) == 0
        else:
            g = 1 if kind == "rms_gemma" else 0
            assert lib().hwocr_add_rmsnorm(None, 0, 0, 0, None, p(x), D, p(w), p(xn), D, None, rows, D, 1e-6, g, st()) == 0
            assert lib().hwocr_rmsnorm_fp8(p(x), D, p(w), p(q), p(s), D, rows, D, 1e-6, g, st()) == 0
        sync()
        q2, s2 = _quant_gpu(xn)
        assert torch.equal(q, q2) and torch.equal(s, s2), kind


def test_attn_prefill_hd256_long_reads():
    """The Gemma-prefill kernel at 8 reads (XCD-dealt grid), 8 query heads on one KV head, several K/V tiles and a ragged tail."""
    hd, Hq, lens = 256, 8, ATTN_HD256_LENS
    nseg, Lp = len(lens), 704
    q = randbf(nseg, Lp, Hq, hd, seed=41)
    k = randbf(nseg, 1, Lp, hd, seed=42)
    v = randbf(nseg, 1, Lp, hd, seed=43)
    vt = v.transpose(2, 3).contiguous()
    for s, n in enumerate(lens):
        vt[s, :, :, n:] = float("nan")
        k[s, :, n:, :] = 1e4
    out = torch.zeros(nseg, Lp, Hq * hd, dtype=torch.bfloat16, device=DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    scale = hd ** -0.5
    rc = lib().hwocr_attn_prefill(p(q), p(k), p(vt), p(out), p(lens_d), nseg, Hq, Hq, hd, max(lens), 0,
                                  Lp * Hq * hd, hd, Hq * hd, Lp * hd, Lp * hd, hd, hd * Lp, hd * Lp, Lp, Lp * Hq * hd, Hq * hd,
                                  scale, 0, st())
    assert rc == 0
    sync()
    for s, n in enumerate(lens):
        want = _sdpa_ref(q[s, :n].float().permute(1, 0, 2), k[s, :, :n].float(), v[s, :, :n].float(), False, scale)
        assert_close_bf16(out[s, :n].view(n, Hq, hd), want, ulps=4.0, atol=4e-3, what=f"attn hd256 read {s}")
    assert torch.isfinite(out.float()).all()
