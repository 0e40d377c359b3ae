"""The engine against the CPU oracle at FULL WIDTHS (missing #3): Qwen2-VL-2B, Qwen2.5-VL-7B (olmOCR-2) and PaliGemma-3B
shapes with every width, head count, vocabulary and the real page size (1008 x 1008 / 896 x 896) as benchmarked, but two tower
blocks and two decoder layers, so that oracle/ (the torch-CPU restatement that is bit-identical to HF's bf16 classes on the tiny
goldens, tests/test_oracle_model.py) finishes in seconds.  What the tiny goldens cannot reach runs here against the oracle and
not only against itself: 5184- / 4096-token attention segments, the 256 x 256 GEMM tiles of every tower / prefill Linear, head_dim
80 / 72-padded / 256, the 151 936- / 257 216-wide LM head, M-RoPE over a 36 x 36 image grid.

Tolerances = tests/test_model_gpu.py's: teacher-forced logits mean-abs <= 5e-3 x scale, 99.9 % within 3e-2 x scale,
none beyond 6e-2 x scale (scale = max(1, max |logit|)), top-1 agreement on every step whose oracle margin exceeds 0.05; image
embeddings within 2 bf16 ulps of the tensor's scale."""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

pytestmark = pytest.mark.gpu

N_NEW = 6


def _check_logits(got, want, toks_engine, toks_oracle, what):
    got, want = got.float().cpu(), want.float()
    scale = max(1.0, float(want.abs().max()))
    d = (got - want).abs()
    assert float(d.mean()) <= 5e-3 * scale, f"{what}: mean logit error {float(d.mean())} (scale {scale})"
    assert float(d.flatten().quantile(0.999)) <= 3e-2 * scale, f"{what}: p99.9 {float(d.flatten().quantile(0.999))} (scale {scale})"
    assert float(d.max()) <= 6e-2 * scale, f"{what}: teacher-forced logits differ by {float(d.max())} (scale {scale})"
    top2 = want.topk(2, -1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.05
    agree = torch.tensor([a == b for a, b in zip(toks_engine, toks_oracle)])
    assert bool(agree[decisive].all()), (what, toks_engine, toks_oracle)


def _check_embeddings(eng, page, want, what):
    emb, _, tok_rows = eng.encode_pages([page])
    torch.cuda.synchronize()
    got = emb[torch.from_numpy(tok_rows[0]).long().to(emb.device)].float().cpu()
    want = want.float()
    assert got.shape == want.shape
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= 2 * 2 ** -7 * scale, f"{what}: image embeddings differ by {err} (scale {scale})"


# (the third case: an ODD number of decoder layers — the <= 16-read decode step alternates the residual stream between two buffers
# per layer, csrc/runtime.hip — and a single tower block)
@pytest.mark.parametrize("preset,over", [("qwen2-vl-2b", {}), ("qwen2.5-vl-7b", {"fullatt": (1,)}), ("qwen2-vl-2b", {"depth": 1, "layers": 3})])
def test_qwen_full_width_two_layers_against_the_oracle(preset, over):
    from handwritten_ocr_amd import engine, imageproc, preprocess, synth, tokenizer
    from handwritten_ocr_amd.compat import config
    from oracle import image_ref
    from oracle.qwen2vl_ref import Qwen2VLRef, RefConfig, rope_index

    cfg = dataclasses.replace(engine.preset(preset), **{"depth": 2, "layers": 2, **over})
    sd = engine.random_state_dict(cfg, seed=0, device="cuda")
    eng = engine.ReadEngine(cfg, sd, max_reads=4, ctx=2048, vit_batch=2, prefill_batch=2)
    rc = RefConfig(depth=cfg.depth, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, family=cfg.family,
                   vit_inter=cfg.vit_inter, window_size=cfg.window_size, fullatt=tuple(cfg.fullatt), hidden=cfg.hidden,
                   layers=cfg.layers, q_heads=cfg.q_heads, kv_heads=cfg.kv_heads, inter=cfg.inter, vocab=cfg.vocab, tie=cfg.tie,
                   image_token_id=cfg.image_token_id, vision_start_id=cfg.vision_start_id, vision_end_id=cfg.vision_end_id,
                   eos_ids=tuple(cfg.eos_ids), pad_id=cfg.pad_id)
    ref = Qwen2VLRef(rc, {k: v.cpu() for k, v in sd.items()})
    del sd
    raw = Image.fromarray(synth.make_page(7, 1024, 1024), "RGB")
    strategies = [config.PREPROCESSING_STRATEGIES[0], config.PREPROCESSING_STRATEGIES[2]]   # two different images
    imgs = [preprocess.apply_strategy(raw, s, quiet=True) for s in strategies]
    pages = [imageproc.prepare_page(im, cfg.patch_size, cfg.merge, config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS) for im in imgs]
    assert pages[0].shape == (1008, 1008, 3) and not np.array_equal(pages[0], pages[1])
    proc = tokenizer.Processor(cfg, tokenizer.ByteTokenizer(cfg, fold_unknown=True))
    prompts = [proc.chat_ids(config.OCR_PROMPT, proc.image_tokens(p)) for p in pages]
    embed = ref.w("model.language_model.embed_tokens.weight")
    want_logits, want_toks = [], []
    with torch.no_grad():
        for r, im in enumerate(imgs):
            pv, grid = image_ref.pixel_values(im, config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS)
            ids = torch.from_numpy(np.asarray(prompts[r])).long()
            img = ref.vision(torch.from_numpy(pv), [grid])
            if r == 0:
                _check_embeddings(eng, pages[0], img, f"{preset} tower")
            x = F.embedding(ids, embed).clone()
            x[ids == cfg.image_token_id] = img.to(x.dtype)
            pos3, delta = rope_index(ids, cfg.image_token_id, [grid], cfg.merge)
            cache = [None] * cfg.layers
            last = ref.lm_head(ref.decoder(x, pos3, cache)[-1:])[0]   # the LM head on the last prompt row only (HF computes it on
            steps, toks = [], []                                        # every row and keeps this one, generation/utils.py:2894)
            for n in range(N_NEW):
                lf = last.float().clone()
                lf[list(cfg.eos_ids)] = -float("inf")                   # min_new == max_new
                steps.append(last)
                toks.append(int(torch.argmax(lf)))
                if n + 1 < N_NEW:
                    last = ref.step(toks[-1], cache, delta)
            want_logits.append(torch.stack(steps))
            want_toks.append(toks)
    toks, logits = eng.generate(pages, prompts, max_new=N_NEW, min_new=N_NEW, forced=np.asarray(want_toks), return_logits=True)
    for r in range(len(pages)):
        _check_logits(logits[r], want_logits[r], toks[r], want_toks[r], f"{preset} read {r}")
    eng.close()


def test_paligemma_full_width_two_layers_against_the_oracle():
    from handwritten_ocr_amd import engine, imageproc, synth
    from oracle.paligemma_ref import PaliGemmaRef, PaliRefConfig

    cfg = dataclasses.replace(engine.preset("paligemma-3b"), depth=2, layers=2)
    sd = engine.random_state_dict(cfg, seed=0, device="cuda")
    eng = engine.ReadEngine(cfg, sd, max_reads=4, ctx=4352, vit_batch=2, prefill_batch=2)
    rc = PaliRefConfig(v_layers=cfg.depth, v_hidden=cfg.embed_dim, v_heads=cfg.num_heads, v_inter=cfg.vit_inter, patch_size=cfg.patch_size,
                       image_size=cfg.image_size, hidden=cfg.hidden, layers=cfg.layers, q_heads=cfg.q_heads, kv_heads=cfg.kv_heads,
                       head_dim=cfg.head_dim, inter=cfg.inter, vocab=cfg.vocab, rope_theta=cfg.rope_theta,
                       image_token_id=cfg.image_token_id, eos_ids=tuple(cfg.eos_ids), pad_id=cfg.pad_id)
    ref = PaliGemmaRef(rc, {k: v.cpu() for k, v in sd.items()})
    del sd
    lut = imageproc.pixel_lut((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    rng = np.random.default_rng(0)
    want_logits, want_toks, pages, prompts = [], [], [], []
    with torch.no_grad():
        for r in range(2):
            page = imageproc.prepare_square(Image.fromarray(synth.make_page(11 + r, 1024, 1024), "RGB"), cfg.image_size)
            pv = torch.from_numpy(np.stack([lut[c][page[:, :, c]] for c in range(3)]))
            n_img = (cfg.image_size // cfg.patch_size) ** 2
            ids_np = np.asarray([cfg.image_token_id] * n_img + [cfg.bos_id] + rng.integers(3, 1000, size=12 + 4 * r).tolist(), np.int32)
            ids = torch.from_numpy(ids_np).long()
            img = ref.vision(pv)
            if r == 0:
                _check_embeddings(eng, page, img, "paligemma-3b tower")
            mask = ids == cfg.image_token_id
            x = ref.embed(torch.where(mask, torch.zeros_like(ids), ids))
            x[mask] = img.to(x.dtype)
            cache = [None] * cfg.layers
            last = ref.lm_head(ref.decoder(x, torch.arange(len(ids)) + 1, cache, bidirectional=True)[-1:])[0]
            steps, toks = [], []
            for n in range(N_NEW):
                lf = last.float().clone()
                lf[list(cfg.eos_ids)] = -float("inf")
                steps.append(last)
                toks.append(int(torch.argmax(lf)))
                if n + 1 < N_NEW:
                    last = ref.step(toks[-1], cache)
            want_logits.append(torch.stack(steps))
            want_toks.append(toks)
            pages.append(page)
            prompts.append(ids_np)
    toks, logits = eng.generate(pages, prompts, max_new=N_NEW, min_new=N_NEW, forced=np.asarray(want_toks), return_logits=True)
    for r in range(2):
        _check_logits(logits[r], want_logits[r], toks[r], want_toks[r], f"paligemma-3b read {r}")
    eng.close()


def test_qwen2vl_2b_full_depth_against_the_oracle():
    """BASELINE config 2's model at FULL depth too (32 tower blocks, 28 decoder layers: 2.2 B parameters, the bench's weights): one
    read of a 1008 x 1008 page, teacher-forced, against the oracle's full-depth read (≈10-15 s of host time — bench.py's
    cpu_baseline leg is this very computation, so the test reuses it).  Same tolerances as the depth-2 cases."""
    import bench
    from handwritten_ocr_amd import engine, synth
    from handwritten_ocr_amd.compat import config

    cfg = engine.preset("qwen2-vl-2b")
    sd = engine.random_state_dict(cfg, seed=0, device="cuda")
    eng = engine.ReadEngine(cfg, sd, max_reads=4, ctx=2048, vit_batch=2, prefill_batch=2)
    sd_cpu = {k: v.to("cpu") for k, v in engine.normalize_keys(sd).items()}
    del sd
    raw = np.ascontiguousarray(synth.make_page(3, 1024, 1024))
    _, ora = bench.cpu_baseline(cfg, sd_cpu, raw, config.PREPROCESSING_STRATEGIES[2], 512, 3, n_dec=5)
    del sd_cpu
    r = bench.full_depth_parity(eng, ora)
    eng.close()
    assert r["steps"] == 6
    assert r["mean_abs_err_over_scale"] <= 5e-3 and r["p999_abs_err_over_scale"] <= 3e-2 and r["max_abs_err_over_scale"] <= 6e-2, r
    assert r["top1_agreement_on_decisive_steps"] == 1.0, r
