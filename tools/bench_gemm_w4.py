#!/usr/bin/env python3
"""The four-wave (128 x 128 per wave) experiment of csrc/gemm256w4.hip (diagnostic build) against the product kernel and the vendor
library on the same operands: correctness (against torch's fp32 product of the bf16 operands) and TFLOP/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

_lib._build.use_diag_library()
hip, p = _lib.hip(), _lib.ptr
dbg = C.CDLL(_lib._build.HIP_LIB)
dbg.hwocr_debug_gemm_w4.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]
for name, M, N, K in (("check", 1024, 768, 512), ("vit fc1", 62208, 5120, 1280), ("vit fc2", 62208, 1280, 5120), ("vit qkv", 62208, 3840, 1280),
                      ("8k cube", 8192, 8192, 8192), ("16k x 8k", 16384, 8192, 8192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    o4 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    o8 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    st = _lib.stream_handle()
    run4 = lambda: dbg.hwocr_debug_gemm_w4(p(x), p(w), p(b), p(o4), M, N, K, st)  # noqa: E731
    run8 = lambda: hip.hwocr_gemm_wide(p(x), p(w), p(b), None, p(o8), M, N, K, K, K, N, 0, 0, st)  # noqa: E731
    runl = lambda: torch.nn.functional.linear(x, w)  # noqa: E731
    assert run4() == 0 and (M < 1024 or run8() == 0)
    torch.cuda.synchronize()
    if M * N <= 1 << 24:
        ref = (x.float() @ w.float().t() + b.float())
        err = (o4.float() - ref).abs().max().item()
        print(f"{name}: max |w4 - fp32 reference| = {err:.4f} (bf16 ulp at the output scale ~ {ref.abs().max().item() / 256:.4f}); "
              f"w4 == product kernel bit for bit: {bool(torch.equal(o4, o8))}", flush=True)
    else:
        same = bool(torch.equal(o4, o8))
        print(f"{name}: w4 == product kernel bit for bit: {same}", flush=True)
    res = []
    for fn in (run4, run8, runl):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(2.0 * M * N * K / ms / 1e9)
    run4()
    torch.cuda.synchronize()
    nwg = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
    buf = (C.c_ulonglong * (2 * nwg))()
    dbg.hwocr_debug_gemm_w4_stamps.argtypes = [C.c_void_p, C.c_int]
    if dbg.hwocr_debug_gemm_w4_stamps(buf, 2 * nwg) == 0:
        import numpy as np
        v = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 2).astype(np.float64)
        cyc, ticks = v[:, 0].mean(), v[:, 1].mean()
        print(f"          four-wave main loop: {cyc / (K // 64):7.0f} shader cycles per K tile (2048 = the matrix pipe's), clock {cyc * 100 / ticks:5.0f} MHz",
              flush=True)
    print(f"{name:9s} M={M} N={N} K={K}:  four-wave {res[0]:7.1f}   product (8 waves) {res[1]:7.1f}   library {res[2]:7.1f}  TFLOP/s", flush=True)
