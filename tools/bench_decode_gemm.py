#!/usr/bin/env python3
"""Micro-benchmark of the decode-step GEMMs (M = reads in flight): hwocr_gemm_skinny (fragment-tiled weights) against
hwocr_gemm_wide at the same shapes.  Reports the weight-streaming rate (N*K*2 bytes / time).  Buffers rotate over
several copies of W so that the 256 MB infinity cache cannot serve the weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

lib = _lib.hip()
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 126
SHAPES = [("2b qkv", 2048, 1536, 5), ("2b o", 1536, 1536, 5), ("2b gate_up", 17920, 1536, 4), ("2b down", 1536, 8960, 5),
          ("7b qkv", 4608, 3584, 5), ("7b o", 3584, 3584, 5), ("7b gate_up", 37888, 3584, 4), ("7b down", 3584, 18944, 5)]
g = torch.Generator(device=dev).manual_seed(0)
st = _lib.stream_handle()


def pick_splitk(K, N, want=400):
    chunks = (K + 255) // 256
    tiles = (N + 31) // 32
    s = max(1, min(chunks, (want + tiles - 1) // tiles))
    per = (chunks + s - 1) // s
    return (chunks + per - 1) // per


def timeit(fn, reps=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N, K, epi in ([] if (len(sys.argv) > 2 and sys.argv[2] == "only") else SHAPES):
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    x = torch.randn(B, K, device=dev, generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(ncopy)]
    wt = []
    for w in ws:
        t = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
        assert lib.hwocr_tile_weights(_lib.ptr(w), _lib.ptr(t), N, K, K, st) == 0
        wt.append(t)
    no = N // 2 if epi == 4 else N
    sk = 1 if epi == 4 else pick_splitk(K, N)
    out = torch.empty(max(sk, 1) * B * no, dtype=torch.float32, device=dev)
    outw = torch.empty(B, no, dtype=torch.bfloat16, device=dev)

    def skinny(i):
        rc = lib.hwocr_gemm_skinny(_lib.ptr(x), _lib.ptr(wt[i % ncopy]), None, _lib.ptr(out), B, N, K, K, K, no, epi, sk, 1, st)
        assert rc == 0, rc

    def wide(i):
        rc = lib.hwocr_gemm_wide(_lib.ptr(x), _lib.ptr(ws[i % ncopy]), None, None, _lib.ptr(outw), B, N, K, K, K, no, 0,
                                 4 if epi == 4 else 0, st)
        assert rc == 0, rc

    us_s, us_w = timeit(skinny), timeit(wide)
    gb = N * K * 2 / 1e9
    print(f"{name:11s} B={B} N={N:6d} K={K:6d} splitk={sk:2d}  skinny {us_s:7.1f} us {gb / us_s * 1e3:6.2f} TB/s   "
          f"wide(no split-K) {us_w:7.1f} us {gb / us_w * 1e3:6.2f} TB/s", flush=True)

# ---- split-K sweep for the slab-producing GEMMs, with the slab-summing consumer (add_rmsnorm) timed behind each
if len(sys.argv) > 2 and sys.argv[2] == "first":
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] in ("sweep", "only"):
    print("split-K sweep: GEMM us + add_rmsnorm us (rows = B)")
    for name, N, K, epi in SHAPES:
        if epi != 5 and False:
            continue
        ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
        x = torch.randn(B, K, device=dev, generator=g).to(torch.bfloat16)
        wt = []
        for _ in range(ncopy):
            w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
            t = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
            assert lib.hwocr_tile_weights(_lib.ptr(w), _lib.ptr(t), N, K, K, st) == 0
            wt.append(t)
        h = torch.randn(B, N, device=dev, generator=g).to(torch.bfloat16)
        nw = torch.ones(N, device=dev, dtype=torch.bfloat16)
        hn = torch.empty(B, N, device=dev, dtype=torch.bfloat16)
        res = []
        for sk in ((1,) if epi == 4 else (1, 2, 3, 4, 6, 7, 8, 10, 12, 14, 16, 20, 24)):
            kt = K // 64
            per = (kt + sk - 1) // sk
            if (sk - 1) * per >= kt:
                continue
            out = torch.empty(sk * B * N, dtype=torch.float32, device=dev)
            no = N // 2 if epi == 4 else N

            def gemm(i):
                assert lib.hwocr_gemm_skinny(_lib.ptr(x), _lib.ptr(wt[i % ncopy]), None, _lib.ptr(out), B, N, K, K, K, no, epi, sk, 1, st) == 0

            def both(i):
                gemm(i)
                if N > 4096 or epi == 4:
                    return
                assert lib.hwocr_add_rmsnorm(_lib.ptr(out), sk, B * N, N, None, _lib.ptr(h), N, _lib.ptr(nw), _lib.ptr(hn), N, None, B, N,
                                             1e-6, 0, st) == 0

            tg, tb = timeit(gemm), timeit(both)
            res.append(f"s={sk}: {tg:5.1f}+{tb - tg:4.1f}={tb:5.1f}")
        print(f"{name:9s} N={N} K={K}: " + "  ".join(res), flush=True)
