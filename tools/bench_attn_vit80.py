"""Time the ViT attention at the page shape of the bench (12 pages x 16 heads x 5184 tokens x head_dim 80, the tower's buffer
layout).  HWOCR_ATTN_SLACK / HWOCR_VIT80_KERNEL (x | 12 | 4) select the variant.  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import _lib  # noqa: E402

nimg, P, heads, hd = 12, 5184, 16, 80
rows, DH = nimg * P, heads * hd
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.randn(heads, rows, hd, generator=g).to(torch.bfloat16).cuda()
k = torch.randn(heads, rows, hd, generator=g).to(torch.bfloat16).cuda()
vt = torch.randn(heads, hd, rows + 64, generator=g)[:, :, :rows].contiguous().to(torch.bfloat16).cuda()
out = torch.zeros(rows, DH, dtype=torch.bfloat16, device="cuda")
lens = torch.full((nimg,), P, dtype=torch.int32, device="cuda")
lib, p = _lib.hip(), _lib.ptr


def run():
    rc = lib.hwocr_attn_prefill(p(q), p(k), p(vt), p(out), p(lens), nimg, heads, 1, hd, P, 0, P * hd, rows * hd, hd, P * hd, rows * hd,
                                hd, P, hd * rows, rows, P * DH, DH, (hd ** -0.5), 0, _lib.stream_handle())
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
fl = 4.0 * P * P * hd * heads * nimg
print(f"kernel={os.environ.get('HWOCR_VIT80_KERNEL', 'default')} slack={os.environ.get('HWOCR_ATTN_SLACK', 'default')}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s  checksum {float(out.float().abs().mean()):.6f}")
