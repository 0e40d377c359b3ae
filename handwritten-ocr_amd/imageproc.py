"""Page -> pixels for the vision tower: the host half of the Qwen2-VL image processor.

Mirrors what `processor.apply_chat_template(...)` does to the image at ocr_agent/tools.py:756-762 through
HF `Qwen2VLImageProcessorPil` (image_processing_pil_qwen2_vl.py:57-83 smart_resize, :126-150 resize, :226-229
rescale + normalize).  Only the geometry (RGB convert + bicubic resize, PIL) stays on the host; rescale, normalise,
patchify and the bf16 cast run in the `hwocr_patchify` kernel from the 3x256-entry table built here with the
library's own arithmetic (float64 v/255 -> float32, then float32 (x - mean) / std).
"""
from __future__ import annotations

import math

import numpy as np
from PIL import Image

IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)   # HF utils/constants.py:5-6 (OPENAI_CLIP_MEAN / STD)
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> tuple[int, int]:
    """Target (h, w): multiples of `factor`, area within [min_pixels, max_pixels], aspect kept as closely as possible."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h = round(height / factor) * factor
    w = round(width / factor) * factor
    if h * w > max_pixels:
        shrink = math.sqrt((height * width) / max_pixels)
        h = max(factor, math.floor(height / shrink / factor) * factor)
        w = max(factor, math.floor(width / shrink / factor) * factor)
    elif h * w < min_pixels:
        grow = math.sqrt(min_pixels / (height * width))
        h = math.ceil(height * grow / factor) * factor
        w = math.ceil(width * grow / factor) * factor
    return h, w


def prepare_page(img: Image.Image, patch: int, merge: int, min_pixels: int, max_pixels: int) -> np.ndarray:
    """PIL page -> uint8 [H', W', 3] at the resolution the vision tower sees."""
    if img.mode != "RGB":
        img = img.convert("RGB")
    h, w = smart_resize(img.height, img.width, patch * merge, min_pixels, max_pixels)
    if (h, w) != (img.height, img.width):
        img = img.resize((w, h), resample=Image.BICUBIC, reducing_gap=None)
    return np.asarray(img, dtype=np.uint8)


def prepare_square(img: Image.Image, size: int) -> np.ndarray:
    """PIL page -> uint8 [size, size, 3]: the SigLIP processor's plain bicubic resize, aspect not kept
    (HF siglip/image_processing_pil_siglip.py: size {height, width}, resample BICUBIC)."""
    if img.mode != "RGB":
        img = img.convert("RGB")
    if (img.height, img.width) != (size, size):
        img = img.resize((size, size), resample=Image.BICUBIC, reducing_gap=None)
    return np.asarray(img, dtype=np.uint8)


def pixel_lut(image_mean=IMAGE_MEAN, image_std=IMAGE_STD) -> np.ndarray:
    """float32 [3][256]: the normalised value of every (channel, byte).  SigLIP / PaliGemma: mean = std = 0.5."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    mean = np.array(image_mean, dtype=np.float32)
    std = np.array(image_std, dtype=np.float32)
    return ((v[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)


def vision_positions(gh: int, gw: int, merge: int) -> tuple[np.ndarray, np.ndarray]:
    """(row, col) of every patch in the processor's merge-block-major order (HF vision_utils.py:81-127)."""
    hp = np.broadcast_to(np.arange(gh, dtype=np.int32)[:, None], (gh, gw))
    wp = np.broadcast_to(np.arange(gw, dtype=np.int32)[None, :], (gh, gw))

    def blocks(t):
        return t.reshape(gh // merge, merge, gw // merge, merge).transpose(0, 2, 1, 3).reshape(-1)

    return blocks(hp), blocks(wp)


def window_order(gh: int, gw: int, merge: int, window_size: int, patch: int) -> tuple[np.ndarray, np.ndarray]:
    """Window partition of the Qwen2.5-VL tower for one gh x gw patch grid (HF vision_utils.py:130-188).

    Returns (order, win_lens): `order[j]` = index (row-major over the merged gh/merge x gw/merge grid) of the merged
    token at position j once tokens are regrouped window by window (windows row-major, tokens row-major inside a
    window); `win_lens[w]` = PATCHES (merge^2 per merged token) in the w-th non-empty window.  The library pads the
    grid by `side - n % side` on each axis — a whole empty window row/column when n is already a multiple — and drops
    the empty windows again, so only non-empty ones are listed."""
    side = window_size // merge // patch
    lh, lw = gh // merge, gw // merge
    idx = np.arange(lh * lw, dtype=np.int32).reshape(lh, lw)
    order, lens = [], []
    for wy in range(0, lh, side):
        for wx in range(0, lw, side):
            tile = idx[wy: wy + side, wx: wx + side].reshape(-1)
            order.append(tile)
            lens.append(tile.size * merge * merge)
    return np.concatenate(order), np.asarray(lens, dtype=np.int32)


def mrope_positions(ids: np.ndarray, image_token_id: int, grids: list[tuple[int, int, int]], merge: int):
    """3-axis decoder positions of one prompt + the offset generated tokens continue from.
    HF modeling_qwen2_vl.py:944-1058: text runs count up on all axes; an image run gets (t, h, w) grid coordinates
    offset by the running position, which then advances by max(h, w) / merge."""
    T = len(ids)
    pos = np.zeros((3, T), dtype=np.int32)
    is_img = ids == image_token_id
    cur = 0
    i = 0
    g = iter(grids)
    while i < T:
        j = i
        while j < T and is_img[j] == is_img[i]:
            j += 1
        n = j - i
        if not is_img[i]:
            pos[:, i:j] = cur + np.arange(n, dtype=np.int32)
            cur += n
        else:
            t, h, w = next(g)
            lh, lw = h // merge, w // merge
            if t * lh * lw != n:
                raise ValueError(f"image placeholder run of {n} tokens does not match grid {t}x{h}x{w}")
            tt, hh, ww = np.meshgrid(np.arange(t), np.arange(lh), np.arange(lw), indexing="ij")
            pos[0, i:j] = tt.reshape(-1) + cur
            pos[1, i:j] = hh.reshape(-1) + cur
            pos[2, i:j] = ww.reshape(-1) + cur
            cur += max(h, w) // merge
        i = j
    return pos, int(pos.max()) + 1 - T
