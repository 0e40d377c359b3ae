"""How far the E4M3 wide GEMMs move the engine from the HF bf16 goldens (tiny PaliGemma / Qwen2-VL models): prints the
statistics tests/test_model_paligemma_gpu.py::test_fp8_* bound.  Run on the GPU box: python tools/diag_fp8.py"""
import os
import sys

import numpy as np
import torch
from PIL import Image
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import engine, imageproc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def run(fp8):
    sd = load_file(os.path.join(GOLD, "paligemma_tiny_weights.safetensors"))
    g = load_file(os.path.join(GOLD, "paligemma_tiny_bf16.safetensors"))
    eng = engine.ReadEngine(engine.preset("tinypg"), sd, max_reads=8, ctx=256, vit_batch=2, prefill_batch=2, fp8=fp8)
    for c in ("a", "b"):
        page = imageproc.prepare_square(Image.fromarray(g[f"{c}.page"].numpy(), "RGB"), eng.cfg.image_size)
        emb, grids, rows = eng.encode_pages([page])
        torch.cuda.synchronize()
        want = g[f"{c}.projector"].float()
        got = emb[torch.from_numpy(rows[0]).long().to(emb.device)].float().cpu()
        sc = float(want.abs().max())
        print(f"fp8={fp8} case {c}: projector max err / scale = {float((got - want).abs().max()) / sc:.4f}  mean = {float((got - want).abs().mean()) / sc:.5f}")
        forced = g[f"{c}.greedy_tokens"].numpy()[None]
        n = forced.shape[1]
        toks, logits = eng.generate([page], [g[f"{c}.input_ids"].numpy()], max_new=n, min_new=n, forced=forced, return_logits=True)
        want = g[f"{c}.step_logits"].float()
        d = (logits[0].float().cpu() - want).abs()
        scale = max(1.0, float(want.abs().max()))
        top2 = want.topk(2, -1).values
        margin = top2[:, 0] - top2[:, 1]
        agree = np.array([a == b for a, b in zip(toks[0], forced[0].tolist())])
        print(f"   logits: scale {scale:.3f} mean {float(d.mean()) / scale:.5f} p99.9 {float(d.flatten().quantile(0.999)) / scale:.4f} "
              f"max {float(d.max()) / scale:.4f}; agree {int(agree.sum())}/{n}; smallest margin among agreeing "
              f"{float(margin[torch.from_numpy(agree)].min()):.4f}; largest margin among disagreeing "
              f"{float(margin[torch.from_numpy(~agree)].max()) if (~agree).any() else 0.0:.4f}")
    eng.close()


if __name__ == "__main__":
    run(False)
    run(True)
