#!/usr/bin/env python3
"""Where a wave of attention_vit80x.hip spends its cycles: builds that unit alone with -DVIT80X_STAMPS into a scratch library (s_memtime
stamps around the phases; the stamps themselves cost a few percent), runs the bench shape (12 pages x 16 heads x 5184 tokens) and prints
cycles per tile and per MFMA for: DMA issue + glue, phase A (QK^T beside the weights), phase B (PV beside the maximum), wait + barrier.
Run on the GPU box.  argv[1] (optional) = output directory for the scratch library (default /tmp)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import build  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
so = os.path.join(out_dir, "libvit80x_dbg.so")
extra = os.environ.get("VIT80X_EXTRA", "").split()
subprocess.run([build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value",
                "-DVIT80X_STAMPS", *extra, *build.EXTRA_FLAGS["attention_vit80x.hip"], "-I" + build.INCLUDE, "-I" + build.CSRC,
                os.path.join(build.CSRC, "attention_vit80x.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.vit80x_debug.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                                                        ctypes.c_void_p]
nimg, P, heads, hd = 12, 5184, 16, 80
rows = nimg * P
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.randn(heads, rows, hd, generator=g).to(torch.bfloat16).cuda()
k = torch.randn(heads, rows, hd, generator=g).to(torch.bfloat16).cuda()
vt = torch.randn(heads, hd, rows + 64, generator=g)[:, :, :rows].contiguous().to(torch.bfloat16).cuda()
out = torch.zeros(rows, heads * hd, dtype=torch.bfloat16, device="cuda")
lens = torch.full((nimg,), P, dtype=torch.int32, device="cuda")
stamps = torch.zeros(64 * 4 * 6, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.vit80x_debug(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), lens.data_ptr(), nimg, heads, P, rows, hd ** -0.5,
                          float(os.environ.get("HWOCR_ATTN_SLACK", "8")), stamps.data_ptr(), st)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
s6 = stamps.view(64, 4, 6).double().cpu()
s = s6[:, :, :4]
nt = P // 64
print(f"{ms:.3f} ms per launch (with stamps)  checksum {float(out.float().abs().mean()):.6f}")
for name, i, mf in (("DMA issue + glue", 0, 0), ("phase A", 1, 20), ("phase B", 2, 24), ("rescale + wait + barrier", 3, 0)):
    per_tile = s[:, :, i].mean() / nt
    print(f"  {name:26s} {per_tile:8.0f} cycles per tile" + (f"  = {per_tile / mf:5.1f} per MFMA" if mf else "")
          + f"   (waves: min {s[:, :, i].min() / nt:.0f} max {s[:, :, i].max() / nt:.0f})")
print(f"  total {s.sum(-1).mean() / nt:8.0f} cycles per tile = {s.sum(-1).mean() / nt / 44:.1f} per MFMA")
print(f"  shader clock inside the tile loop: {float((s6[:, :, 4] / s6[:, :, 5]).mean()) * 0.1:.2f} GHz (s_memtime cycles per 100 MHz tick)")
