#!/usr/bin/env python3
"""Time hwocr_layernorm at the tower shape (12 pages: 62208 rows x 1280) - a read + write streaming kernel.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

lib, p, st = _lib.hip(), _lib.ptr, _lib.stream_handle()
rows, D = 62208, 1280
xs = [torch.randn(rows, D, device="cuda").to(torch.bfloat16) for _ in range(4)]  # rotate: 4 x 159 MB > the 256 MB Infinity Cache
w, b = torch.randn(D, device="cuda").to(torch.bfloat16), torch.randn(D, device="cuda").to(torch.bfloat16)
out = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")


def run(i):
    assert lib.hwocr_layernorm(p(xs[i % 4]), p(w), p(b), p(out), rows, D, D, D, 1e-6, st) == 0


for i in range(4):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40):
    run(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
print(f"layernorm {rows} x {D}: {us:.1f} us  {2 * rows * D * 2 / us / 1e6:.2f} TB/s  checksum {float(out.float().abs().mean()):.5f}")
