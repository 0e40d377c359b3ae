#!/usr/bin/env python3
"""Where a tile of the 256x256 GEMM spends its time: HWOCR_GEMM_ABLATE=10 makes wave 0 of every workgroup stamp the 100 MHz
clock at the start of each tile's main loop, at its end and after the epilogue (EPI_LINEAR bf16 instance).
    HWOCR_GEMM_ABLATE=10 python tools/bench_gemm_timeline.py [M N K]
HWOCR_TL_CUS=<n>: the launch goes into a stream masked to n CUs (hwocr_stream_create_cumask) with the persistent grid sized to it:
what the shader clock and the time per K tile do when fewer CUs draw power."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handwritten_ocr_amd import _lib  # noqa: E402

assert os.environ.get("HWOCR_GEMM_ABLATE") == "10", "run with HWOCR_GEMM_ABLATE=10"
_lib._build.use_diag_library()  # the -DHWOCR_DIAG build (csrc/diag/): the shipped library has no ablation / timeline variants
lib = C.CDLL(_lib._build.HIP_LIB)
hip = _lib.hip()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (62208, 5120, 1280)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
cus = int(os.environ.get("HWOCR_TL_CUS", "256"))
if cus < 256:
    words = (C.c_uint * 8)(*[sum(1 << b for b in range(32) if 256 - cus <= 32 * wd + b < 256) for wd in range(8)])
    h = C.c_void_p()
    _lib.check(hip.hwocr_stream_create_cumask(words, 8, C.byref(h)))
    torch.cuda.set_stream(torch.cuda.ExternalStream(h.value))
    hip.hwocr_set_cu_budget(cus)
st = _lib.stream_handle()
for _ in range(8):
    assert hip.hwocr_gemm_wide(_lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(out), M, N, K, K, K, N, 0, 0, st) == 0
torch.cuda.synchronize()
n = 256 * 64 * 4
buf = (C.c_ulonglong * n)()
lib.hwocr_debug_gemm_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.hwocr_debug_gemm_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 4).astype(np.float64) * 0.01  # us
tiles = (M + 255) // 256 * ((N + 255) // 256)
per = [len(range(b, tiles, cus)) for b in range(cus)]
loop, epi, gap, first = [], [], [], []
for b in range(cus):
    for i in range(per[b]):
        loop.append(t[b, i, 1] - t[b, i, 0])
        epi.append(t[b, i, 2] - t[b, i, 1])
        if i + 1 < per[b]:
            gap.append(t[b, i + 1, 0] - t[b, i, 2])
t0 = min(t[b, 0, 0] for b in range(cus))
t1 = max(t[b, per[b] - 1, 2] for b in range(cus) if per[b])
print(f"M={M} N={N} K={K} on {cus} CUs: {tiles} tiles, {K // 64} K tiles each; kernel span {t1 - t0:.1f} us")
for name, v in (("main loop", loop), ("epilogue (incl. next prologue issue)", epi), ("inter-tile wait + barrier", gap)):
    v = np.asarray(v)
    print(f"  {name:38s} mean {v.mean():6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}")
print(f"  first main-loop start spread over workgroups: {max(t[b, 0, 0] for b in range(cus)) - t0:.2f} us; "
      f"last tile end spread: {t1 - min(t[b, per[b] - 1, 2] for b in range(cus) if per[b]):.2f} us")
raw = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 4)
mhz = [(float(raw[b, per[b] - 1, 3]) - float(raw[b, 0, 3])) / ((float(raw[b, per[b] - 1, 0]) - float(raw[b, 0, 0])) * 0.01)
       for b in range(cus) if per[b] > 1]
print(f"  shader clock while the kernel runs (s_memtime ticks per wall-clock us): mean {np.mean(mhz):.0f} MHz")
ideal = 2.0 * 256 * 256 * K / (2.5e15 / 256) * 1e6
print(f"  MFMA time of one tile at the 2.5 PF peak: {ideal:.2f} us")

# ---- what a work queue could buy: every workgroup's own mean time per tile (loop + epilogue + hand-over), and the same tiles dealt
# to the same workgroups (a) round-robin as the kernel does, (b) from one queue in the order workgroups come free (list scheduling)
per_tile = np.array([(t[b, per[b] - 1, 2] - t[b, 0, 0]) / per[b] if per[b] else np.nan for b in range(cus)])
ok = ~np.isnan(per_tile)
print(f"  time per tile by workgroup: mean {np.nanmean(per_tile):.2f} us, min {np.nanmin(per_tile):.2f}, p10 {np.nanpercentile(per_tile, 10):.2f}, "
      f"p90 {np.nanpercentile(per_tile, 90):.2f}, max {np.nanmax(per_tile):.2f}")
for x in range(8):
    v = per_tile[x::8]
    print(f"    workgroups = {x} mod 8: {np.nanmean(v):6.2f} us per tile (min {np.nanmin(v):.2f}, max {np.nanmax(v):.2f})")
static = max(per[b] * per_tile[b] for b in range(cus) if per[b])
import heapq
heap = [(0.0, b) for b in range(cus) if ok[b]]
heapq.heapify(heap)
end = 0.0
for _ in range(tiles):
    at, b = heapq.heappop(heap)
    at += per_tile[b]
    end = max(end, at)
    heapq.heappush(heap, (at, b))
print(f"  span with each workgroup's own mean tile time: round-robin {static:.1f} us, one queue {end:.1f} us ({100 * (1 - end / static):.1f} % less); "
      f"work / workgroups = {tiles * np.nanmean(per_tile) / ok.sum():.1f} us")
