"""Image half of the processor: product (handwritten_ocr_amd.imageproc) and oracle (oracle/image_ref.py) against
outputs of HF's Qwen2VLImageProcessorPil (tests/golden/image_kats.*).  Exact."""
import numpy as np
import pytest
from PIL import Image
from safetensors.torch import load_file

from handwritten_ocr_amd import imageproc
from oracle import image_ref
from tests._golden import GOLD, load_json

META = load_json("image_kats.json")
T = load_file(f"{GOLD}/image_kats.safetensors")


@pytest.mark.parametrize("fn", [imageproc.smart_resize, image_ref.smart_resize], ids=["product", "oracle"])
def test_smart_resize_table(fn):
    for row in META["smart_resize"]:
        try:
            out = list(fn(row["h"], row["w"], 28, row["min_pixels"], row["max_pixels"]))
        except ValueError:
            out = "ValueError"
        assert out == row["out"], row


def test_oracle_pixel_values_equal_hf():
    for i, m in enumerate(META["images"]):
        pv, grid = image_ref.pixel_values(Image.fromarray(T[f"img{i}.page"].numpy(), "RGB"), m["min_pixels"], m["max_pixels"])
        assert list(grid) == m["grid_thw"]
        assert np.array_equal(pv, T[f"img{i}.pixel_values"].numpy())


def test_product_resize_and_lut_equal_hf():
    lut = imageproc.pixel_lut()
    for i, m in enumerate(META["images"]):
        page = imageproc.prepare_page(Image.fromarray(T[f"img{i}.page"].numpy(), "RGB"), 14, 2, m["min_pixels"], m["max_pixels"])
        gh, gw = page.shape[0] // 14, page.shape[1] // 14
        chw = np.stack([lut[c][page[:, :, c]] for c in range(3)])
        x = chw.reshape(3, gh // 2, 2, 14, gw // 2, 2, 14).transpose(1, 4, 2, 5, 0, 3, 6)
        x = np.broadcast_to(x[:, :, :, :, :, None], (*x.shape[:5], 2, 14, 14)).reshape(gh * gw, 1176)
        assert np.array_equal(x, T[f"img{i}.pixel_values"].numpy())


def test_positions_match_oracle():
    import torch
    from oracle.qwen2vl_ref import rope_index, vision_position_ids

    ph, pw = imageproc.vision_positions(8, 12, 2)
    want = vision_position_ids(8, 12, 2)
    assert np.array_equal(ph, want[:, 0].numpy()) and np.array_equal(pw, want[:, 1].numpy())
    ids = np.array([3, 4, 9, 9, 9, 9, 9, 9, 7, 8, 9, 9, 1], dtype=np.int32)
    pos, delta = imageproc.mrope_positions(ids, 9, [(1, 4, 6), (1, 2, 4)], 2)
    wpos, wdelta = rope_index(torch.from_numpy(ids).long(), 9, [(1, 4, 6), (1, 2, 4)], 2)
    assert np.array_equal(pos, wpos.numpy()) and delta == wdelta
    with pytest.raises(ValueError):
        imageproc.mrope_positions(ids, 9, [(1, 4, 4), (1, 2, 4)], 2)


def test_siglip_square_resize_and_lut_match_hf():
    """PaliGemma / SigLIP preprocessing: plain bicubic resize to the tower size, (x/255 - 0.5)/0.5 — against the HF
    SiglipImageProcessorPil output stored with the PaliGemma goldens."""
    import os

    import torch
    from safetensors.torch import load_file

    from handwritten_ocr_amd import imageproc
    from tests._golden import GOLD
    from PIL import Image

    g = load_file(os.path.join(GOLD, "paligemma_tiny_fp32.safetensors"))
    lut = imageproc.pixel_lut((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    for case in ("a", "b"):
        page = imageproc.prepare_square(Image.fromarray(g[f"{case}.page"].numpy(), "RGB"), 56)
        got = np.stack([lut[c][page[:, :, c]] for c in range(3)])
        assert np.array_equal(got, g[f"{case}.pixel_values"].numpy())
