"""Host side of the device preprocessing: the restated Pillow coefficient tables (gpupre.bicubic_coeffs) drive a numpy copy
of Resample.c's two passes to the very pixels Image.resize(BICUBIC) returns; chain coverage rules."""
import numpy as np
import pytest
from PIL import Image

from handwritten_ocr_amd import gpupre


def _resample(img, n_out, axis):
    b, k = gpupre.bicubic_coeffs(img.shape[axis], n_out)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((n_out,) + src.shape[1:], np.uint8)
    for o in range(n_out):
        x0, n = b[o]
        acc = (src[x0:x0 + n] * k[o, :n].astype(np.int64).reshape((-1,) + (1,) * (src.ndim - 1))).sum(0) + (1 << 21)
        out[o] = np.clip(acc >> 22, 0, 255)
    return np.moveaxis(out, 0, axis)


@pytest.mark.parametrize("H,W,oh,ow", [(64, 80, 56, 84), (200, 120, 196, 112), (90, 90, 252, 252), (512, 512, 504, 504)])
def test_coefficient_tables_reproduce_pillow_bicubic(H, W, oh, ow):
    img = np.random.default_rng(H * W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), resample=Image.BICUBIC, reducing_gap=None))
    got = _resample(_resample(img, ow, 1), oh, 0)  # horizontal pass first, uint8 in between (Resample.c ImagingResample)
    assert np.array_equal(got, want)


def test_identity_size_is_identity():
    b, k = gpupre.bicubic_coeffs(50, 50)
    assert all(int(k[o, : b[o, 1]].sum()) == 1 << 22 for o in range(50))
    img = np.random.default_rng(3).integers(0, 256, (50, 7, 3), dtype=np.uint8)
    assert np.array_equal(_resample(img, 50, 0), img)


def test_chain_coverage_rules():
    from handwritten_ocr_amd import preprocess
    from handwritten_ocr_amd.compat import config

    if preprocess._cv2() is not None:
        assert not gpupre.supported(["high_contrast"])
        return
    for s in config.PREPROCESSING_STRATEGIES:
        assert gpupre.supported(s)
    assert gpupre.supported("original") and gpupre.supported([])
    assert not gpupre.supported(["binarize", "sharpen"])  # PIL would sharpen a mode-L image
    assert gpupre.supported(["no_such_transform", "high_contrast"])  # unknown names are skipped (tools.py:661-664)
