#!/usr/bin/env python3
"""Is the wide GEMM bound by the number of CUs or by the chip's power budget?  The same launch on CU-masked streams of 256 / 224 /
192 / 160 / 128 CUs (hwocr_stream_create_cumask; masks take the same number of CUs from every XCD and shader engine) with the
persistent grid sized to the partition (hwocr_set_cu_budget): TFLOP/s and TFLOP/s per CU."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

lib, p = _lib.hip(), _lib.ptr
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def masked(n):
    words = (C.c_uint * 8)(*[sum(1 << b for b in range(32) if ncu - n <= 32 * w + b < ncu) for w in range(8)])
    h = C.c_void_p()
    _lib.check(lib.hwocr_stream_create_cumask(words, 8, C.byref(h)), "hwocr_stream_create_cumask")
    return torch.cuda.ExternalStream(h.value), h


for name, M, N, K, epi in (("vit fc1", 62208, 5120, 1280, 2), ("vit fc2", 62208, 1280, 5120, 0), ("8k cube", 8192, 8192, 8192, 0)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for n in (256, 224, 192, 160, 128):
        s, h = masked(n)
        lib.hwocr_set_cu_budget(n)
        with torch.cuda.stream(s):
            run = lambda: _lib.check(lib.hwocr_gemm_wide(p(x), p(w), p(b), None, p(out), M, N, K, K, K, N, 0, epi, _lib.stream_handle()))  # noqa: E731
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            s.synchronize()
        ms = e0.elapsed_time(e1) / 20
        tf = 2.0 * M * N * K / ms / 1e9
        print(f"{name:8s} {n:3d} CUs  {ms:7.3f} ms  {tf:7.1f} TFLOP/s  {tf / n:5.2f} per CU", flush=True)
        lib.hwocr_set_cu_budget(0)
        torch.cuda.synchronize()
        _lib.check(lib.hwocr_stream_destroy(h))
