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


def test_a_cu_masked_stream_runs_where_its_mask_says_and_computes_the_same():
    """hwocr_stream_create_cumask (tools/bench_partition.py, DESIGN.md §3 - measured, not the default schedule): a prefix of 64 mask
    bits is 8 CUs of every XCD, the complement the other 24; a persistent GEMM launched into the masked stream with the matching CU
    budget returns the bytes of the same GEMM on the whole chip."""
    import ctypes as C

    from handwritten_ocr_amd import _lib

    lib = _lib.hip()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count

    def stream(bits):
        words = (C.c_uint * 8)(*[sum(1 << b for b in range(32) if 32 * w + b in bits) for w in range(8)])
        h = C.c_void_p()
        _lib.check(lib.hwocr_stream_create_cumask(words, 8, C.byref(h)), "hwocr_stream_create_cumask")
        return torch.cuda.ExternalStream(h.value), h

    def where(s):
        out = torch.zeros(4096, 2, dtype=torch.int32, device="cuda")
        with torch.cuda.stream(s):
            _lib.check(lib.hwocr_probe_placement(_lib.ptr(out), 4096, 40000, _lib.stream_handle()), "hwocr_probe_placement")
            s.synchronize()
        o = out.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        per = {}
        for x, c in zip((o[:, 0] & 0xF).tolist(), ((o[:, 1] >> 8) & 0xFF).tolist()):  # XCC_ID; HW_ID cu [11:8] sh [12] se [15:13]
            per.setdefault(x, set()).add(c)
        return {x: len(v) for x, v in per.items()}

    small, hs = stream(set(range(64)))
    big, hb = stream(set(range(64, ncu)))
    try:
        # (on the boxes of this round: exactly 8 / 24 CUs of every XCD; which physical CUs a part has fused off may shift a few)
        ws, wb = where(small), where(big)
        assert sum(ws.values()) == 64 and sorted(ws) == list(range(8)) and all(4 <= v <= 12 for v in ws.values()), ws
        assert sum(wb.values()) == ncu - 64 and sorted(wb) == list(range(8)) and all(16 <= v <= 32 for v in wb.values()), wb
        g = torch.Generator(device="cuda").manual_seed(3)
        M, N, K = 4096, 1280, 1280
        X = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
        want, got = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.hwocr_gemm_wide(_lib.ptr(X), _lib.ptr(W), None, None, _lib.ptr(want), M, N, K, K, K, N, 0, 0, _lib.stream_handle()))
        torch.cuda.synchronize()
        assert lib.hwocr_set_cu_budget(ncu - 64) == 0
        try:
            with torch.cuda.stream(big):
                _lib.check(lib.hwocr_gemm_wide(_lib.ptr(X), _lib.ptr(W), None, None, _lib.ptr(got), M, N, K, K, K, N, 0, 0, _lib.stream_handle()))
                big.synchronize()
        finally:
            lib.hwocr_set_cu_budget(0)
        assert torch.equal(got, want)
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.hwocr_stream_destroy(hs), "hwocr_stream_destroy")
        _lib.check(lib.hwocr_stream_destroy(hb), "hwocr_stream_destroy")
