"""PaliGemma (BASELINE config 4) on the MI355X: SigLIP tower with heads zero-padded 72 -> 80, Gemma decoder with head_dim 256,
bidirectional prompt prefix — the read engine against the outputs of the real HF classes (tests/golden/paligemma_tiny_*)
and the CPU oracle.  Tolerances as tests/test_model_gpu.py."""
import os

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

from tests._golden import GOLD, load_json  # noqa: E402


def _gold(tag="bf16"):
    return load_file(os.path.join(GOLD, f"paligemma_tiny_{tag}.safetensors"))


@pytest.fixture(scope="module")
def eng():
    from handwritten_ocr_amd import engine

    sd = load_file(os.path.join(GOLD, "paligemma_tiny_weights.safetensors"))
    e = engine.ReadEngine(engine.preset("tinypg"), sd, max_reads=8, ctx=256, vit_batch=2, prefill_batch=2)
    yield e
    e.close()


def _page(eng, g, case):
    from handwritten_ocr_amd import imageproc

    return imageproc.prepare_square(Image.fromarray(g[f"{case}.page"].numpy(), "RGB"), eng.cfg.image_size)


def test_siglip_tower_matches_hf(eng):
    g = _gold()
    for case in ("a", "b"):
        emb, grids, tok_rows = eng.encode_pages([_page(eng, g, case)])
        torch.cuda.synchronize()
        want = g[f"{case}.projector"].float()
        got = emb[torch.from_numpy(tok_rows[0]).long().to(emb.device)].float().cpu()
        assert got.shape == want.shape
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2 * 2 ** -7 * scale, float((got - want).abs().max())


@pytest.mark.parametrize("batched", [False, True])
def test_teacher_forced_logits_match_hf(eng, batched):
    g = _gold()
    n = load_json("paligemma_tiny.json")["cases"]["a"]["n_new"]
    cases = ["a", "b"] if batched else ["a"]
    pages = [_page(eng, g, c) for c in cases]
    prompts = [g[f"{c}.input_ids"].numpy() for c in cases]
    forced = np.stack([g[f"{c}.greedy_tokens"].numpy() for c in cases])
    toks, logits = eng.generate(pages, prompts, max_new=n, min_new=n, forced=forced, return_logits=True)
    for r, c in enumerate(cases):
        want = g[f"{c}.step_logits"].float()
        d = (logits[r].float().cpu() - want).abs()
        scale = max(1.0, float(want.abs().max()))
        assert float(d.mean()) <= 5e-3 * scale, f"case {c}: mean logit error {float(d.mean())} (scale {scale})"
        assert float(d.flatten().quantile(0.999)) <= 3e-2 * scale, f"case {c}: p99.9 {float(d.flatten().quantile(0.999))}"
        assert float(d.max()) <= 6e-2 * scale, f"case {c}: {float(d.max())} (scale {scale})"
        top2 = want.topk(2, -1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 0.05
        agree = torch.tensor([a == b for a, b in zip(toks[r], g[f"{c}.greedy_tokens"].tolist())])
        assert bool(agree[decisive].all())


def test_graph_decode_equals_eager(eng):
    g = _gold()
    cases = ["a", "b", "a"]
    pages = [_page(eng, g, c) for c in cases]
    prompts = [g[f"{c}.input_ids"].numpy() for c in cases]
    eager = eng.generate(pages, prompts, max_new=12, min_new=12, use_graph=False)
    graph1 = eng.generate(pages, prompts, max_new=12, min_new=12, use_graph=True)
    graph2 = eng.generate(pages, prompts, max_new=12, min_new=12, use_graph=True)
    assert eager == graph1 == graph2 and eager[0] == eager[2]
