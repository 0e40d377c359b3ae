"""BASELINE configs 1 and 4 at their stated sizes on the MI355X (neither shape was ever executed).

  config 1  "single 512x512 sample page, tiny random-init ViT-S + 125M decoder": the `small` preset, one page -> 504x504, P = 1296
            patches, M = 324 image tokens, the three strategy reads.
  config 4  "PaliGemma-3B / SigLIP-So400m encoder, 896x896 tiles, fp8 MFMA": the `paligemma-3b` preset with fp8=True, one
            1024x1024 page resized to 896x896 (4096 image tokens), the three strategy reads.
No oracle finishes these widths in seconds and HF has no fp8 path, so — as tests/test_fullsize_gpu.py does for config 2 — the
checks are the size-independent properties of the read path; the arithmetic itself is pinned on the tiny goldens
(tests/test_model_gpu.py, tests/test_model_paligemma_gpu.py) and kernel by kernel at these widths (tests/test_ops_gpu.py
DECODE_GEMM_SHAPES has both presets' decode GEMMs).  The fp8 leg is a stated tolerance, parity unpinned."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

CASES = {
    "config1_small_512": dict(preset="small", side=512, fp8=False, ctx=1024, want_hw=(504, 504), want_tokens=324),
    "config4_paligemma3b_896_fp8": dict(preset="paligemma-3b", side=1024, fp8=True, ctx=4352, want_hw=(896, 896), want_tokens=4096),
}


@pytest.fixture(scope="module", params=list(CASES))
def setup(request):
    from handwritten_ocr_amd import engine, imageproc, preprocess, synth, tokenizer
    from handwritten_ocr_amd.compat import config

    c = CASES[request.param]
    cfg = engine.preset(c["preset"])
    sd = engine.random_state_dict(cfg, seed=0, device="cuda")
    eng = engine.ReadEngine(cfg, sd, max_reads=6, ctx=c["ctx"], vit_batch=3, prefill_batch=3, fp8=c["fp8"])
    del sd
    img = Image.fromarray(synth.make_page(7, c["side"], c["side"]), "RGB")
    pages = []
    for s in config.PREPROCESSING_STRATEGIES[:3]:
        pre = preprocess.apply_strategy(img, s, quiet=True)
        pages.append(imageproc.prepare_square(pre, cfg.image_size) if cfg.family == "paligemma" else
                     imageproc.prepare_page(pre, cfg.patch_size, cfg.merge, config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS))
    proc = tokenizer.Processor(cfg, tokenizer.ByteTokenizer(cfg, fold_unknown=True))
    prompts = [proc.chat_ids(config.OCR_PROMPT, proc.image_tokens(p)) for p in pages]
    yield c, eng, pages, prompts
    eng.close()
    del eng
    torch.cuda.empty_cache()


def test_shapes_of_the_config(setup):
    c, eng, pages, prompts = setup
    assert pages[0].shape[:2] == c["want_hw"]
    assert int((prompts[0] == eng.cfg.image_token_id).sum()) == c["want_tokens"]
    assert not np.array_equal(pages[0], pages[2]), "a different strategy must change the pixels"


def test_reads_are_independent_of_batch_position_and_company(setup):
    c, eng, pages, prompts = setup
    n = 12
    together = eng.generate(pages, prompts, max_new=n, min_new=n)
    assert all(len(t) == n for t in together)
    assert eng.generate(pages[::-1], prompts[::-1], max_new=n, min_new=n)[::-1] == together
    alone = [eng.generate([p], [q], max_new=n, min_new=n)[0] for p, q in zip(pages, prompts)]
    assert alone == together
    assert int(eng.lens[0]) == len(prompts[-1]) + n and int(eng.n_gen[0]) == n


def test_graph_replay_equals_eager_and_duplicate_reads_agree(setup):
    c, eng, pages, prompts = setup
    n = 10
    dup_pages, dup_prompts = [pages[0], pages[2], pages[0]], [prompts[0], prompts[2], prompts[0]]
    eager = eng.generate(dup_pages, dup_prompts, max_new=n, min_new=n, use_graph=False)
    graph = eng.generate(dup_pages, dup_prompts, max_new=n, min_new=n, use_graph=True)
    assert eager == graph and eager[0] == eager[2]
    lg = eng._bufs["logits"][:3].float()
    assert torch.isfinite(lg).all()
    assert torch.equal(lg[0], lg[2]) and not torch.equal(lg[0], lg[1])


def test_continuous_batching_equals_lockstep(setup):
    """More reads than decode slots: generate_stream refills slots as reads finish; same tokens as the lockstep batch."""
    c, eng, pages, prompts = setup
    n = 8
    want = eng.generate(pages, prompts, max_new=n, min_new=n)
    many_pages, many_prompts = pages * 3, prompts * 3   # 9 reads through 6 slots
    got = eng.generate_stream(many_pages, many_prompts, max_new=n, min_new=n, sync_every=4)
    assert got == want * 3
