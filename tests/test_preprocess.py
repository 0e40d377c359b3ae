"""Preprocessing strategies against the reference's own outputs (tests/golden/preprocess_kats.json, produced by
ocr_agent.tools.preprocess_image in an image without OpenCV -> PIL fallbacks).  Pixel-exact."""
import contextlib
import hashlib
import io
import os

import numpy as np
import pytest
from PIL import Image

from handwritten_ocr_amd import preprocess
from tests._golden import load_json
from handwritten_ocr_amd.synth import make_page

K = load_json("preprocess_kats.json")


@pytest.mark.skipif(preprocess._cv2() is not None, reason="goldens pin the no-OpenCV fallbacks")
def test_strategies_pixel_exact(tmp_path):
    assert K["cv2_available"] is False
    pages = {}
    for c in K["cases"]:
        key = (c["seed"], c["h"], c["w"], c["mode"])
        if key not in pages:
            img = Image.fromarray(make_page(c["seed"], c["h"], c["w"]), "RGB")
            if c["mode"] == "L":
                img = img.convert("L")
            path = tmp_path / f"page{c['seed']}.png"
            img.save(path)
            pages[key] = str(path)
        src = pages[key]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = preprocess.preprocess_image(src, c["strategy"])
        assert (out == src) == c["returns_input_path"]
        assert buf.getvalue() == c["stdout"]
        res = Image.open(out)
        assert res.mode == c["out_mode"] and list(res.size) == c["out_size"]
        assert hashlib.sha256(np.asarray(res).tobytes()).hexdigest() == c["pixels_sha256"], c["strategy"]
        if out != src:
            assert os.path.basename(out).startswith(c["basename_starts"]) and out.endswith(c["suffix"])
            # the in-memory chain is the same transform without the file round trip
            mem = preprocess.apply_strategy(Image.open(src), c["strategy"], quiet=True)
            assert np.array_equal(np.asarray(mem), np.asarray(res))
            os.unlink(out)


def test_strategy_table_matches_reference_config():
    from handwritten_ocr_amd.compat import config

    assert [list(s) for s in config.PREPROCESSING_STRATEGIES] == K["strategies_config"]
