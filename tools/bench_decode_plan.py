#!/usr/bin/env python3
"""Time the five GEMMs of one decode step exactly as hwocr_decode_step issues them (shapes, epilogues and split-K from the
library's own launch planner, engine.decode_plan) on the MI355X: microseconds per launch with the weights rotated over enough
copies that neither L2 nor the Infinity Cache can serve them, and the bytes each launch brings into the compute units.

    python tools/bench_decode_plan.py [preset] [reads]            e.g.  qwen2-vl-2b 252
Kernel experiment switches are environment variables read by the library (HWOCR_STREAM_*): run one process per setting."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib, engine  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-2b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 252
lib = _lib.hip()
dev = "cuda"
st = _lib.stream_handle()
g = torch.Generator(device=dev).manual_seed(0)
plan = engine.decode_plan(engine.preset(preset), B)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("HWOCR_"))
print(f"# {preset} reads={B} {tag}")
total = 0.0
for name in engine.DECODE_GEMMS:
    N, K, epi, sk, variant = plan[name]
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    x = torch.randn(B, K, device=dev, generator=g).to(torch.bfloat16)
    wt = []
    for _ in range(ncopy):
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        t = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
        assert lib.hwocr_tile_weights(_lib.ptr(w), _lib.ptr(t), N, K, K, st) == 0
        wt.append(t)
        del w
    no = N // 2 if epi in (4, 7) else N
    out = torch.empty(max(sk, 1) * B * no, dtype=torch.float32, device=dev)

    def run(i):
        assert lib.hwocr_gemm_skinny(_lib.ptr(x), _lib.ptr(wt[i % ncopy]), None, _lib.ptr(out), B, N, K, K, K, no, epi, sk, 1, st) == 0

    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    per_layer = us * (1 if name == "lm_head" else 1)
    total += us if name != "lm_head" else 0.0
    print(f"{name:8s} N={N:6d} K={K:5d} epi={epi} splitk={sk:2d} {variant:48s} {us:7.1f} us  W {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del wt
print(f"per layer (qkv+o+gate_up+down): {total:.1f} us")
