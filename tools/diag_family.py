"""Diagnostic (GPU box): per-stage differences engine vs HF goldens for both tiny families."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from handwritten_ocr_amd import engine, imageproc
from tests import _golden

for fam, preset in (("qwen2_vl", "tiny"), ("qwen2_5_vl", "tiny25")):
    cfg = engine.preset(preset)
    eng = engine.ReadEngine(cfg, _golden.tiny_weights(torch.bfloat16, fam), max_reads=8, ctx=256, vit_batch=2, prefill_batch=2)
    g = _golden.tiny_case("bf16", fam)
    meta = _golden.tiny_meta(fam)["cases"]
    for case in ("a", "b"):
        page = imageproc.prepare_page(Image.fromarray(g[f"{case}.page"].numpy(), "RGB"), cfg.patch_size, cfg.merge, cfg.min_pixels, cfg.max_pixels)
        emb, grids, rows = eng.encode_pages([page])
        torch.cuda.synchronize()
        want = g[f"{case}.merger"].float()
        got = emb[torch.from_numpy(rows[0]).long().to(emb.device)].float().cpu()
        d = (got - want).abs()
        print(fam, case, "merger maxdiff %.4f scale %.3f  mean %.5f" % (d.max(), want.abs().max(), d.mean()))
        n = meta[case]["n_new"]
        forced = g[f"{case}.greedy_tokens"].numpy()[None]
        toks, logits = eng.generate([page], [g[f"{case}.input_ids"].numpy()], max_new=n, min_new=n, forced=forced, return_logits=True)
        w = g[f"{case}.step_logits"].float()
        dd = (logits[0].float().cpu() - w).abs()
        print(fam, case, "step logits maxdiff per step", [round(float(x), 3) for x in dd.max(-1).values], "scale %.3f mean %.5f" % (w.abs().max(), dd.mean()))
    eng.close()
