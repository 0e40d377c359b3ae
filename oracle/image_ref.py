"""ORACLE — test infrastructure, not product code.

numpy restatement of the image half of the reference's processor call (ocr_agent/tools.py:756-762 ->
HF Qwen2VLImageProcessorPil): smart_resize (image_processing_pil_qwen2_vl.py:57-83), PIL bicubic resize (:126-150,
image_transforms.py:367), float64 rescale cast to float32 (image_transforms.py:118-122), float32 normalise
(:419-439), patchify (image_processing_pil_qwen2_vl.py:152-187).  Pinned by tests/golden/image_kats.*.
"""
import math

import numpy as np
from PIL import Image

MEAN = [0.48145466, 0.4578275, 0.40821073]
STD = [0.26862954, 0.26130258, 0.27577711]


def smart_resize(h, w, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280):
    if max(h, w) / min(h, w) > 200:
        raise ValueError("aspect ratio")
    hb, wb = round(h / factor) * factor, round(w / factor) * factor
    if hb * wb > max_pixels:
        beta = math.sqrt((h * w) / max_pixels)
        hb, wb = max(factor, math.floor(h / beta / factor) * factor), max(factor, math.floor(w / beta / factor) * factor)
    elif hb * wb < min_pixels:
        beta = math.sqrt(min_pixels / (h * w))
        hb, wb = math.ceil(h * beta / factor) * factor, math.ceil(w * beta / factor) * factor
    return hb, wb


def pixel_values(img: Image.Image, min_pixels, max_pixels, patch=14, merge=2, tps=2):
    """PIL page -> (float32 [P, 3*tps*patch*patch], (1, gh, gw))."""
    img = img.convert("RGB")
    h, w = smart_resize(img.height, img.width, patch * merge, min_pixels, max_pixels)
    arr = np.asarray(img.resize((w, h), resample=Image.BICUBIC)).transpose(2, 0, 1)       # C,H,W uint8
    x = (arr.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = ((x.T - np.array(MEAN, np.float32)) / np.array(STD, np.float32)).T
    gh, gw = h // patch, w // patch
    x = x.reshape(3, gh // merge, merge, patch, gw // merge, merge, patch).transpose(1, 4, 2, 5, 0, 3, 6)
    x = np.broadcast_to(x[:, :, :, :, :, None], (*x.shape[:5], tps, patch, patch)).reshape(gh * gw, 3 * tps * patch * patch)
    return np.ascontiguousarray(x), (1, gh, gw)
