"""Synthetic handwritten pages (no sample image ships with the reference: data/input holds only .gitkeep).

make_page(seed, h, w): bright noisy paper (uint8 U[200,255]), rows of dark pseudo-handwriting stroke polylines
(ink U[0,80], width 2-4 px) and optional ruled lines — the input shape SURVEY.md §8d defines for the pages/sec metric.
"""
from __future__ import annotations

import numpy as np
from PIL import Image, ImageDraw


def make_page(seed: int, h: int = 1024, w: int = 1024, ruled: bool = True) -> np.ndarray:
    rng = np.random.default_rng(seed)
    paper = rng.integers(200, 256, size=(h, w, 1), dtype=np.uint8).repeat(3, axis=2)
    img = Image.fromarray(paper, "RGB")
    d = ImageDraw.Draw(img)
    nlines = max(2, h // 42)
    for i in range(nlines):
        y0 = int((i + 0.6) * h / nlines)
        if ruled:
            d.line([(0, y0 + 6), (w, y0 + 6)], fill=(150, 170, 210), width=1)
        x = int(rng.integers(4, 12))
        while x < w - 8:
            n = int(rng.integers(3, 7))
            pts = [(x + int(rng.integers(0, 10)) + 4 * k, y0 + int(rng.integers(-9, 7))) for k in range(n)]
            ink = int(rng.integers(0, 81))
            d.line(pts, fill=(ink, ink, ink), width=int(rng.integers(2, 5)))
            x += 4 * n + int(rng.integers(6, 18))
    return np.asarray(img)


def tint_page(page: np.ndarray, tint) -> np.ndarray:
    """Paper colour: every channel scaled by tint[c] / 256 in integer arithmetic (the same bytes on any host).  The scribbles of
    make_page differ from page to page only patch by patch; a tint gives a page a GLOBAL look as well, as scanned or
    photographed paper has (tests/golden/trained_*: what the briefly trained tiny checkpoints tell their pages apart by)."""
    t = np.asarray(tint, dtype=np.uint16).reshape(1, 1, 3)
    return ((page.astype(np.uint16) * t) >> 8).astype(np.uint8)
