#!/usr/bin/env python3
"""Time the token-selection kernels at the decode shape (252 reads x vocab 151936 bf16 logits): greedy argmax against the sampling
draw with top-k / top-p on and off.  Run on the GPU box."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

lib, p, st = _lib.hip(), _lib.ptr, _lib.stream_handle()
R, V = int(os.environ.get("READS", 252)), int(os.environ.get("VOCAB", 151936))
i32 = dict(dtype=torch.int32, device="cuda")
logits = (torch.randn(R, V, device="cuda") * 3).to(torch.bfloat16)
cur, lens, fin, ng = torch.zeros(R, **i32), torch.ones(R, **i32), torch.zeros(R, **i32), torch.zeros(R, **i32)
out = torch.zeros(R, 8, **i32)
rid = torch.arange(R, **i32)
eos = (C.c_int * 4)(0, 0, 0, 0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ng.zero_()
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def greedy():
    assert lib.hwocr_argmax_advance(p(logits), V, V, R, p(cur), p(lens), p(ng), p(fin), p(out), 8, 0, eos, 0, 0, None, 1, 1.0, None, st) == 0


print(f"{R} reads x vocab {V}: argmax {timeit(greedy):7.1f} us")
for t, k, tp in ((1.0, 0, 1.0), (0.8, 50, 1.0), (0.8, 0, 0.95), (0.8, 50, 0.95)):
    def draw():
        assert lib.hwocr_sample_advance(p(logits), V, V, R, p(cur), p(lens), p(ng), p(fin), p(out), 8, 0, eos, 0, 0, None, 1, 1.0, t, k, tp, 1,
                                        p(rid), None, st) == 0
    print(f"  draw T={t} top_k={k} top_p={tp}: {timeit(draw):7.1f} us")
