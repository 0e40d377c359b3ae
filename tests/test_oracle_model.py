"""The model oracle (oracle/qwen2vl_ref.py) against outputs of the real HF classes (tests/golden, written by
tools/make_goldens.py).  CPU only."""
import pytest
import torch

from oracle.qwen2vl_ref import Qwen2VLRef, rope_index
from tests._golden import FAMILIES, tiny_case, tiny_meta, tiny_ref_config, tiny_weights


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_matches_hf(tag, dtype, case, family):
    meta = tiny_meta(family)["cases"][case]
    g = tiny_case(tag, family)
    ref = Qwen2VLRef(tiny_ref_config(family), tiny_weights(dtype, family))
    ids = g[f"{case}.input_ids"].long()
    grids = [tuple(meta["grid_thw"])]
    logits, cache, delta = ref.prefill(ids, g[f"{case}.pixel_values"], grids)
    assert delta == meta["rope_delta"]
    assert torch.equal(ref.trace["position_ids"].int(), g[f"{case}.position_ids"])
    # fp32: same ops in the same order -> tight.  bf16: identical module-level rounding points; in the container that
    # wrote the goldens every tensor below is bit-identical to HF's (the bf16 model is loaded with from_pretrained, so
    # its rotary inv_freq buffers stay fp32 as in any real run).  The bound leaves room for another CPU's matmul
    # blocking only: ~2 bf16 ulps of the tensor's scale.
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    for name in ("patch_embed", "vit_block0", "vit_last", "merger", "dec_layer0"):
        want = g[f"{case}.{name}"].float()
        got = ref.trace[name].float()
        assert got.shape == want.shape, name
        scale = float(want.abs().max())
        assert torch.allclose(got, want, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, scale)), \
            f"{name}: max diff {float((got - want).abs().max())} (scale {scale})"
    want = g[f"{case}.prefill_logits"].float()
    assert torch.allclose(logits.float(), want, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(want.abs().max())))


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_oracle_greedy_and_teacher_forced(tag, dtype, family):
    case = "a"
    meta = tiny_meta(family)["cases"][case]
    g = tiny_case(tag, family)
    ref = Qwen2VLRef(tiny_ref_config(family), tiny_weights(dtype, family))
    ids = g[f"{case}.input_ids"].long()
    hf_tokens = g[f"{case}.greedy_tokens"].tolist()
    n = meta["n_new"]
    toks, step_logits = ref.generate(ids, g[f"{case}.pixel_values"], [tuple(meta["grid_thw"])], max_new=n, min_new=n,
                                     forced=hf_tokens)
    want = g[f"{case}.step_logits"].float()
    atol = 1e-4 if dtype == torch.float32 else 3e-2  # bf16 teacher-forced logits max-abs <= 3e-2
    diff = float((step_logits.float() - want).abs().max())
    assert diff <= atol * max(1.0, float(want.abs().max())), diff
    if dtype == torch.float32:
        assert toks == hf_tokens
    else:
        top2 = want.topk(2, dim=-1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 0.05
        agree = torch.tensor([a == b for a, b in zip(toks, hf_tokens)])
        assert bool(agree[decisive].all())


def test_rope_index_text_only_and_two_images():
    ids = torch.tensor([1, 2, 3, 9, 9, 9, 9, 4, 5, 9, 9, 6])
    pos, delta = rope_index(ids, 9, [(1, 4, 4), (1, 2, 4)], 2)
    # text 0..2 ; image 2x2 grid at offset 3 ; text continues at 3 + max(4,4)//2 = 5
    assert pos[:, :3].tolist() == [[0, 1, 2]] * 3
    assert pos[:, 3:7].tolist() == [[3, 3, 3, 3], [3, 3, 4, 4], [3, 4, 3, 4]]
    assert pos[:, 7:9].tolist() == [[5, 6]] * 3
    assert pos[:, 9:11].tolist() == [[7, 7], [7, 7], [7, 8]]
    assert pos[:, 11].tolist() == [9, 9, 9]
    assert delta == 10 - 12


@pytest.mark.parametrize("gh,gw", [(72, 72), (4, 6), (10, 14), (36, 20), (2, 2)])
def test_window_index_partition(gh, gw):
    """Every merged token appears once; windows are contiguous runs; 1008x1008 pages give 81 full 64-patch windows."""
    from oracle.qwen2vl_ref import window_index

    order, lens = window_index(gh, gw, 2, 112, 14)
    n = (gh // 2) * (gw // 2)
    assert sorted(order.tolist()) == list(range(n)) and sum(lens) == gh * gw
    if (gh, gw) == (72, 72):
        assert lens == [64] * 81


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_oracle_repetition_penalty_matches_hf(tag, dtype, family):
    """Greedy with `repetition_penalty` (the deterministic part of the Qwen2.5-VL / olmOCR generation defaults): the token
    stream and the processed scores HF returns, teacher-forced on HF's own stream."""
    for case in ("a", "b"):
        meta = tiny_meta(family)["cases"][case]
        g = tiny_case(tag, family)
        ref = Qwen2VLRef(tiny_ref_config(family), tiny_weights(dtype, family))
        hf = g[f"{case}.rp_tokens"].tolist()
        toks, _ = ref.generate(g[f"{case}.input_ids"].long(), g[f"{case}.pixel_values"], [tuple(meta["grid_thw"])],
                               max_new=len(hf), min_new=len(hf), forced=hf, repetition_penalty=meta["repetition_penalty"])
        got, want = torch.stack(ref.processed_scores), g[f"{case}.rp_scores"]
        finite = torch.isfinite(want)
        assert torch.equal(torch.isfinite(got), finite)
        tol = 1e-4 if dtype == torch.float32 else 1e-2
        assert torch.allclose(got[finite], want[finite], rtol=tol, atol=tol * max(1.0, float(want[finite].abs().max())))
        top2 = want.topk(2, -1).values
        decisive = (top2[:, 0] - top2[:, 1]) > (1e-5 if dtype == torch.float32 else 0.05)
        agree = torch.tensor([a == b for a, b in zip(toks, hf)])
        assert bool(agree[decisive].all())
        if (family, case) == ("qwen2_vl", "a"):  # 10 of its 24 tokens differ from the plain greedy stream
            assert toks != g[f"{case}.greedy_tokens"].tolist(), "the penalty must change this stream"
