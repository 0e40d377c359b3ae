"""The PaliGemma oracle (oracle/paligemma_ref.py: SigLIP tower, projector, Gemma decoder with a bidirectional prefix)
against outputs of the real HF classes (tests/golden/paligemma_tiny_*, written by tools/make_goldens.py).  CPU only.
No product path exists for this family yet; these tests pin the checker the next kernels will be held to."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle.paligemma_ref import PaliGemmaRef, PaliRefConfig
from tests._golden import GOLD, load_json


def _cfg():
    m = load_json("paligemma_tiny.json")["config"]
    v, t = m["vision"], m["text"]
    return PaliRefConfig(v_layers=v["num_hidden_layers"], v_hidden=v["hidden_size"], v_heads=v["num_attention_heads"],
                         v_inter=v["intermediate_size"], patch_size=v["patch_size"], image_size=v["image_size"],
                         hidden=t["hidden_size"], layers=t["num_hidden_layers"], q_heads=t["num_attention_heads"],
                         kv_heads=t["num_key_value_heads"], head_dim=t["head_dim"], inter=t["intermediate_size"],
                         vocab=t["vocab_size"], image_token_id=m["image_token_id"], eos_ids=(m["eos"],), pad_id=m["pad"])


def _ref(dtype):
    sd = load_file(os.path.join(GOLD, "paligemma_tiny_weights.safetensors"))
    return PaliGemmaRef(_cfg(), {k: v.to(dtype) for k, v in sd.items()})


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_matches_hf(tag, dtype, case):
    g = load_file(os.path.join(GOLD, f"paligemma_tiny_{tag}.safetensors"))
    ref = _ref(dtype)
    logits, _ = ref.prefill(g[f"{case}.input_ids"].long(), g[f"{case}.pixel_values"])
    # fp32: same ops in the same order; bf16: same rounding points (bit-identical in the generating container), with room
    # for another CPU's matmul blocking
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    for name in ("patch_embed", "vit_block0", "vit_last", "projector", "dec_layer0"):
        want, got = g[f"{case}.{name}"].float(), ref.trace[name].float()
        assert got.shape == want.shape, name
        scale = max(1.0, float(want.abs().max()))
        assert torch.allclose(got, want, rtol=tol, atol=tol * scale), f"{name}: {float((got - want).abs().max())} (scale {scale})"
    want = g[f"{case}.prefill_logits"].float()
    assert torch.allclose(logits.float(), want, rtol=tol, atol=tol * max(1.0, float(want.abs().max())))


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_oracle_greedy_and_teacher_forced(tag, dtype):
    g = load_file(os.path.join(GOLD, f"paligemma_tiny_{tag}.safetensors"))
    n = load_json("paligemma_tiny.json")["cases"]["a"]["n_new"]
    hf_tokens = g["a.greedy_tokens"].tolist()
    toks, step_logits = _ref(dtype).generate(g["a.input_ids"].long(), g["a.pixel_values"], max_new=n, min_new=n,
                                             forced=hf_tokens)
    want = g["a.step_logits"].float()
    atol = 1e-4 if dtype == torch.float32 else 3e-2
    assert float((step_logits.float() - want).abs().max()) <= atol * max(1.0, float(want.abs().max()))
    top2 = want.topk(2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > (0.0 if dtype == torch.float32 else 0.05)
    agree = torch.tensor([a == b for a, b in zip(toks, hf_tokens)])
    assert bool(agree[decisive].all())
