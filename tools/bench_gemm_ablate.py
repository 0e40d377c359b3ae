"""Where the 256x256 GEMM's time goes: the product kernel against two ablations of itself (HWOCR_GEMM_ABLATE=1: no LDS-DMA after
the prologue — matrix pipe + LDS reads + barriers only; =2: no MFMA — staging + LDS reads + barriers only; =3: 1 without barriers;
=4: 1 without fragment reads; =5: the whole main loop, no epilogue; =6: everything but the epilogue's global stores; =7 / 8 / 9: the epilogue's stores
non-temporal / sc1 / sc0 sc1 instead of the default policy; =11: the same fragments through half as many 32x32x16 MFMAs, WRONG results).  Run on the GPU box once per setting:
HWOCR_GEMM_ABLATE=<0..9> python tools/bench_gemm_ablate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwritten_ocr_amd import _lib  # noqa: E402

_lib._build.use_diag_library()  # the -DHWOCR_DIAG build (csrc/diag/): the shipped library has no ablation variants
lib, p = _lib.hip(), _lib.ptr
for M, N, K in ((62208, 5120, 1280), (62208, 1280, 5120), (62208, 5120, 128), (21248, 17920, 1536), (16384, 8192, 8192)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

    def run():
        assert lib.hwocr_gemm_wide(p(x), p(w), None, None, p(out), M, N, K, K, K, N, 0, 0, _lib.stream_handle()) == 0

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    staged = tiles * 512 * K * 2
    print(f"ablate={os.environ.get('HWOCR_GEMM_ABLATE', '0')} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s-equivalent  "
          f"staged {staged / 1e9:.2f} GB = {staged / ms / 1e9:.2f} TB/s into the CUs")
