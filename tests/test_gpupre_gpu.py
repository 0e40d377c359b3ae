"""Device preprocessing (csrc/imagepre.hip through the C ABI) against Pillow itself and against the host path that
tests/golden/preprocess_kats.json pins: identical pixels."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance, ImageFilter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    from handwritten_ocr_amd import gpupre, preprocess

    if preprocess._cv2() is not None:
        pytest.skip("OpenCV present: the reference takes its cv2 branches, which the device path does not restate")
    return gpupre.StrategyPages()


def _page(h, w, seed):
    from handwritten_ocr_amd import synth

    if min(h, w) >= 128:
        return np.ascontiguousarray(synth.make_page(seed, h=h, w=w))
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("h,w", [(37, 53), (300, 420), (1024, 1024)])
def test_each_transform_equals_pillow(sp, h, w):
    arr = _page(h, w, 5)
    im = Image.fromarray(arr, "RGB")
    dev = torch.from_numpy(arr).cuda()
    hc = sp.high_contrast(dev)
    want_hc = np.asarray(ImageEnhance.Contrast(im).enhance(2.0))
    assert np.array_equal(hc.cpu().numpy(), want_hc)
    want_bin = np.asarray(Image.fromarray(want_hc, "RGB").convert("L").point(lambda v: 255 if v > 128 else 0).convert("RGB"))
    assert np.array_equal(sp.binarize(hc).cpu().numpy(), want_bin)
    want_sh = np.asarray(Image.fromarray(want_hc, "RGB").filter(ImageFilter.SHARPEN))
    assert np.array_equal(sp.sharpen(hc).cpu().numpy(), want_sh)


@pytest.mark.parametrize("lo,n_hi,n", [(100, 32, 64), (100, 31, 64), (100, 33, 64), (0, 1, 2), (254, 1, 2), (17, 4999, 9998),
                                      (17, 4998, 9998), (200, 1 << 19, 1 << 20)])
def test_contrast_mean_is_rounded_on_the_device_as_pillow_rounds_it(sp, lo, n_hi, n):
    """The mean of ImageEnhance.Contrast is int(mean(L) + 0.5); the device takes it from the luma sum without a host round trip
    (hwocr_img_contrast_dev).  Gray pages whose mean sits on, just below and just above a .5 boundary, against Pillow, and the
    host-mean entry point on the same mean: identical bytes."""
    from handwritten_ocr_amd import _lib

    w = 64 if n % 64 == 0 else 2
    g = np.full(n, lo, np.uint8)
    g[:n_hi] = lo + 1
    arr = np.ascontiguousarray(np.repeat(g.reshape(n // w, w, 1), 3, axis=2))
    im = Image.fromarray(arr, "RGB")
    want = np.asarray(ImageEnhance.Contrast(im).enhance(2.0))
    dev = torch.from_numpy(arr).cuda()
    got = sp.high_contrast(dev)
    assert np.array_equal(got.cpu().numpy(), want)
    mean = int(int(sp._sum.item()) / n + 0.5)
    assert mean == int(np.asarray(im.convert("L"), np.int64).sum() / n + 0.5)
    out = torch.empty_like(dev)
    _lib.check(sp.lib.hwocr_img_contrast(_lib.ptr(dev), _lib.ptr(out), 3 * n, mean, 2.0, _lib.stream_handle()), "hwocr_img_contrast")
    assert torch.equal(out, got)


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 56, 84), (300, 420, 280, 392), (1024, 1024, 1008, 1008), (1024, 1024, 896, 896),
                                       (200, 200, 504, 504)])
def test_resize_equals_pillow_bicubic(sp, h, w, oh, ow):
    arr = _page(h, w, 6)
    want = np.asarray(Image.fromarray(arr, "RGB").resize((ow, oh), resample=Image.BICUBIC, reducing_gap=None))
    got = sp.resize(torch.from_numpy(arr).cuda(), oh, ow).cpu().numpy()
    assert np.array_equal(got, want)


def test_strategy_pages_equal_the_host_path(sp):
    """Every configured strategy chain + the image processor's resize: the tensor handed to the tower is the host path's."""
    from handwritten_ocr_amd import engine, imageproc, preprocess
    from handwritten_ocr_amd.compat import config

    c = engine.preset("qwen2-vl-2b")
    arr = _page(1024, 1024, 7)
    im = Image.fromarray(arr, "RGB")
    strategies = list(config.PREPROCESSING_STRATEGIES) + ["original"]
    hw = imageproc.smart_resize(1024, 1024, c.patch_size * c.merge, c.min_pixels, c.max_pixels)
    got = sp.pages(arr, strategies, hw)
    for s, g in zip(strategies, got):
        want = imageproc.prepare_page(preprocess.apply_strategy(im, s, quiet=True), c.patch_size, c.merge, c.min_pixels, c.max_pixels)
        assert np.array_equal(g.cpu().numpy(), want), s


def test_batched_driver_with_device_preprocessing_gives_the_same_states(tmp_path, monkeypatch, capsys):
    """HWOCR_GPU_PREPROCESS=1 in `initial_ocr_batched`: same candidates and merged text as the host preprocessing path (the
    tower sees identical pixels, so the reads are identical token for token)."""
    from handwritten_ocr_amd import batch, preprocess, tools
    from handwritten_ocr_amd.compat import config
    from handwritten_ocr_amd.synth import make_page

    if preprocess._cv2() is not None:
        pytest.skip("OpenCV present")
    monkeypatch.setenv("HWOCR_MODEL", "tiny")
    monkeypatch.setenv("HWOCR_ALLOW_RANDOM_INIT", "1")
    monkeypatch.setenv("HWOCR_MAX_READS", "8")
    monkeypatch.setenv("HWOCR_CTX", "512")
    monkeypatch.setattr(tools, "_ocr_model", None)
    monkeypatch.setattr(tools, "_ocr_processor", None)
    monkeypatch.setattr(config, "OCR_MIN_PIXELS", 28 * 28)
    paths = []
    for i in range(2):
        p = tmp_path / f"p{i}.png"
        Image.fromarray(make_page(80 + i, 70, 100), "RGB").save(p)
        paths.append(str(p))
    # a grey-scale page: the device path only takes plain RGB pages, this one keeps the host path inside the same batch
    p = tmp_path / "p2.png"
    Image.fromarray(make_page(82, 70, 100), "RGB").convert("L").save(p)
    paths.append(str(p))
    params = {"max_new_tokens": 12, "min_new_tokens": 12}
    monkeypatch.setenv("HWOCR_GPU_PREPROCESS", "0")
    host = batch.initial_ocr_batched(paths, params)
    monkeypatch.setenv("HWOCR_GPU_PREPROCESS", "1")
    dev = batch.initial_ocr_batched(paths, params)
    capsys.readouterr()
    assert len(host) == len(dev) == 3
    for a, b in zip(host, dev):
        assert a["current_best"] == b["current_best"]
        assert [c["text"] for c in a["candidates"]] == [c["text"] for c in b["candidates"]]
        assert a["strategies_used"] == b["strategies_used"]
    tools.unload_ocr_model()
