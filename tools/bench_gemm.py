#!/usr/bin/env python3
"""Micro-benchmark of hwocr_gemm_wide at the page-read shapes (MI355X).  TFLOP/s from torch CUDA events.
With `lib` as the first argument the same contraction through torch's library GEMM (hipBLASLt / rocBLAS, no epilogue) is
timed beside it — a yardstick for what a tuned vendor kernel reaches on these shapes, never part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

lib = _lib.hip()
dev = "cuda"
SHAPES = [("vit qkv", 62208, 3840, 1280, 0), ("vit proj", 62208, 1280, 1280, 1), ("vit fc1", 62208, 5120, 1280, 2),
          ("vit fc2", 62208, 1280, 5120, 1), ("dec qkv", 21248, 2048, 1536, 0), ("dec gate_up", 21248, 17920, 1536, 4),
          ("dec down", 21248, 1536, 8960, 1), ("8k cube", 8192, 8192, 8192, 0)]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, epi in SHAPES:
    x = (torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16) if epi != 4 else None
    no = N // 2 if epi == 4 else N
    res = torch.randn(M, no, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, no, device=dev, dtype=torch.bfloat16)
    st = _lib.stream_handle()

    def run():
        rc = lib.hwocr_gemm_wide(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(res) if epi == 1 else None, _lib.ptr(out),
                                 M, N, K, K, K, no, no, epi, st)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    extra = ""
    if len(sys.argv) > 1 and sys.argv[1] == "lib":
        for _ in range(3):
            torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            torch.nn.functional.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        ml = e0.elapsed_time(e1) / reps
        extra = f"   | torch linear (library GEMM, no epilogue) {ml:8.3f} ms {2.0 * M * N * K / ml / 1e9:8.1f} TFLOP/s"
    print(f"{name:12s} M={M:6d} N={N:6d} K={K:5d} epi={epi}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s{extra}", flush=True)
