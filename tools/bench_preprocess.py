"""Host (PIL) against device (csrc/imagepre.hip) strategy preprocessing + resize of 1024x1024 pages, three strategies per page,
as the batched driver runs them.  Run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import engine, gpupre, imageproc, preprocess, synth  # noqa: E402
from handwritten_ocr_amd.compat import config  # noqa: E402

c = engine.preset("qwen2-vl-2b")
strategies = list(config.PREPROCESSING_STRATEGIES)[:3]
pages = [np.ascontiguousarray(synth.make_page(s)) for s in range(8)]
hw = imageproc.smart_resize(1024, 1024, c.patch_size * c.merge, c.min_pixels, c.max_pixels)

t0 = time.perf_counter()
for a in pages:
    im = Image.fromarray(a, "RGB")
    for s in strategies:
        imageproc.prepare_page(preprocess.apply_strategy(im, s, quiet=True), c.patch_size, c.merge, c.min_pixels, c.max_pixels)
host = (time.perf_counter() - t0) / len(pages)

sp = gpupre.StrategyPages()
sp.pages(pages[0], strategies, hw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for a in pages:
    out = sp.pages(a, strategies, hw)
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / len(pages)
print(f"per 1024x1024 page, {len(strategies)} strategies -> {hw}: host (1 core) {host * 1e3:.1f} ms, device incl. upload {dev * 1e3:.2f} ms "
      f"({host / dev:.0f}x); uploads per page: 1 x 3.1 MB instead of {len(strategies)} x {hw[0] * hw[1] * 3 / 1e6:.1f} MB")
