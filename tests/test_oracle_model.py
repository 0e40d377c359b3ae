"""The model oracle (oracle/qwen2vl_ref.py) against outputs of the real HF classes (tests/golden, written by
tools/make_goldens.py).  CPU only."""
import pytest
import torch

from oracle.qwen2vl_ref import Qwen2VLRef, rope_index
from tests._golden import tiny_case, tiny_meta, tiny_ref_config, tiny_weights


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_matches_hf(tag, dtype, case):
    meta = tiny_meta()["cases"][case]
    g = tiny_case(tag)
    ref = Qwen2VLRef(tiny_ref_config(), tiny_weights(dtype))
    ids = g[f"{case}.input_ids"].long()
    grids = [tuple(meta["grid_thw"])]
    logits, cache, delta = ref.prefill(ids, g[f"{case}.pixel_values"], grids)
    assert delta == meta["rope_delta"]
    assert torch.equal(ref.trace["position_ids"].int(), g[f"{case}.position_ids"])
    # fp32: same ops in the same order -> tight; bf16: identical module-level rounding points, so near bit-exact, but
    # SDPA / matmul blocking may differ between call shapes -> 2 bf16 ulps of the tensor's scale
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for name in ("patch_embed", "vit_block0", "vit_last", "merger", "dec_layer0"):
        want = g[f"{case}.{name}"].float()
        got = ref.trace[name].float()
        assert got.shape == want.shape, name
        scale = float(want.abs().max())
        assert torch.allclose(got, want, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, scale)), \
            f"{name}: max diff {float((got - want).abs().max())} (scale {scale})"
    want = g[f"{case}.prefill_logits"].float()
    assert torch.allclose(logits.float(), want, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(want.abs().max())))


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_oracle_greedy_and_teacher_forced(tag, dtype):
    case = "a"
    meta = tiny_meta()["cases"][case]
    g = tiny_case(tag)
    ref = Qwen2VLRef(tiny_ref_config(), tiny_weights(dtype))
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
