"""Two batches in flight on one GPU (pipeline.LanePipeline: batch k's decode beside batch k+1's tower + prefill on two HIP streams,
two host threads) must return exactly what one batch at a time returns — a lane is an ordinary engine over the same weights, and
batches never share state.  Also: an exception in one batch surfaces, and the pipeline is usable afterwards."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from handwritten_ocr_amd import engine, imageproc, pipeline, synth, tokenizer
    from handwritten_ocr_amd.compat import config

    cfg = engine.preset("small")
    sd = engine.random_state_dict(cfg, seed=0, device="cuda")
    eng = engine.ReadEngine(cfg, sd, max_reads=24, ctx=1024, vit_batch=4, prefill_batch=8)
    del sd
    proc = tokenizer.Processor(cfg, tokenizer.ByteTokenizer(cfg, fold_unknown=True))
    batches = []
    for b in range(5):  # batches of different sizes and pages: 24 / 3 / 17 / 24 / 9 reads (both decode paths: <= 16 reads and more)
        n = (24, 3, 17, 24, 9)[b]
        pages = [imageproc.prepare_page(Image.fromarray(synth.make_page(100 * b + i, 512, 512), "RGB"), cfg.patch_size, cfg.merge,
                                        config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS) for i in range(n)]
        prompts = [proc.chat_ids(config.OCR_PROMPT, proc.image_tokens(p)) for p in pages]
        batches.append((pages, prompts))
    pipe = pipeline.LanePipeline(eng, lanes=2)
    yield eng, pipe, batches
    pipe.close()
    eng.close()


@pytest.mark.parametrize("order", ["lockstep", "alternate"])
def test_two_lanes_return_what_one_batch_at_a_time_returns(setup, order):
    eng, pipe, batches = setup
    pipe.order = order   # lockstep: nothing orders the lanes on the device; alternate: tower(k) after prefill(k-1), decode(k) after decode(k-1)
    n = 24
    want = [eng.generate(p, q, max_new=n, min_new=n) for p, q in batches]
    jobs = [(lambda e, hooks, p=p, q=q: e.generate(p, q, max_new=n, min_new=n, hooks=hooks)) for p, q in batches]
    for _ in range(3):   # first pass: lane 1 captures its graphs; later passes replay them
        got = pipe.run(jobs)
        assert got == want
    assert pipe.engines[1].k_cache.data_ptr() != eng.k_cache.data_ptr(), "a lane has its own KV cache"
    assert pipe.engines[1].vit is eng.vit and pipe.engines[1].dec is eng.dec, "and shares the bound weights"
    pipe.order = "lockstep"


def test_ordered_calls_run_in_batch_order(setup):
    eng, pipe, batches = setup
    order = []

    def job(e, hooks, k):
        p, q = batches[k % len(batches)]
        out = e.generate(p[:3], q[:3], max_new=4, min_new=4, hooks=hooks)
        hooks.ordered(lambda: order.append(k))
        return out

    pipe.run([(lambda e, hooks, k=k: job(e, hooks, k)) for k in range(7)])
    assert order == list(range(7))


def test_a_failing_batch_raises_and_the_pipeline_survives(setup):
    eng, pipe, batches = setup
    p, q = batches[1]

    def bad(e, hooks):
        raise ValueError("unreadable page")

    ok = lambda e, hooks: e.generate(p, q, max_new=4, min_new=4, hooks=hooks)  # noqa: E731
    with pytest.raises(ValueError, match="unreadable page"):
        pipe.run([ok, bad, ok, ok])
    torch.cuda.synchronize()
    want = eng.generate(p, q, max_new=4, min_new=4)
    assert pipe.run([ok, ok, ok]) == [want] * 3
