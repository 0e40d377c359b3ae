"""Time the head_dim-256 non-causal prefill attention at the PaliGemma-3B prefill shape (16 reads x 4113 tokens, 8 query heads
on 1 KV head).  HWOCR_HD256_WAVES = 0 (generic kernel) / 4 / 8 selects the kernel.  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import _lib  # noqa: E402

nseg, L, Lp, Hq, hd = 16, 4113, 4160, 8, 256
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.randn(nseg, Lp, Hq, hd, generator=g).to(torch.bfloat16).to(dev)
k = torch.randn(nseg, 1, Lp, hd, generator=g).to(torch.bfloat16).to(dev)
vt = torch.randn(nseg, 1, hd, Lp, generator=g).to(torch.bfloat16).to(dev)
out = torch.zeros(nseg, Lp, Hq * hd, dtype=torch.bfloat16, device=dev)
lens = torch.full((nseg,), L, dtype=torch.int32, device=dev)
lib, p = _lib.hip(), _lib.ptr


def run():
    rc = lib.hwocr_attn_prefill(p(q), p(k), p(vt), p(out), p(lens), nseg, Hq, Hq, hd, L, 0, Lp * Hq * hd, hd, Hq * hd, Lp * hd,
                                Lp * hd, hd, hd * Lp, hd * Lp, Lp, Lp * Hq * hd, Hq * hd, hd ** -0.5, 0, _lib.stream_handle())
    assert rc == 0


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
fl = 4.0 * L * L * hd * Hq * nseg
print(f"HWOCR_HD256_WAVES={os.environ.get('HWOCR_HD256_WAVES', 'default')}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s  checksum {float(out.float().abs().mean()):.6f}")
