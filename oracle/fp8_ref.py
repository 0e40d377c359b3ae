"""ORACLE — test infrastructure, not product code.

CPU restatement of the E4M3 leg of the wide GEMM (include/hwocr.h: hwocr_quant_rows_fp8, hwocr_gemm_wide_fp8).  The
reference has no fp8 path (`run_ocr` loads the checkpoint in bf16, ocr_agent/tools.py:700-709); BASELINE config 4 asks
for "fp8 MFMA on CDNA4", so the contract is this repo's own and is stated here:

  * format: OCP FP8 E4M3 ("e4m3fn": bias 7, no infinities, max 448, subnormals down to 2^-9), round-to-nearest-even —
    pinned in tests/test_oracle_fp8.py against the encodings the OCP 8-bit floating point specification lists;
  * quantisation: one scale per row, scale = max|x| / 448 (1 for an all-zero row), q = e4m3(x * (448 / max|x|));
  * product: fp32 accumulation of the exact fp8 x fp8 products, times (wscale[n] * xscale[m]), then the bf16 epilogue of
    the bf16 GEMM.

What is NOT pinned by any reference output: how far fp8 moves the logits from the bf16 model — tests state the
tolerance they accept against the HF bf16 goldens ("parity unpinned" for the fp8 leg beyond that tolerance).

Only tests/ may import this module.
"""
from __future__ import annotations

import torch

E4M3_MAX = 448.0


def quant_rows(x: torch.Tensor):
    """x [rows][K] (bf16 or fp32) -> (q float8_e4m3fn [rows][K], scale fp32 [rows])."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1)
    nz = amax > 0
    inv = torch.where(nz, torch.tensor(E4M3_MAX) / amax, torch.zeros_like(amax))
    scale = torch.where(nz, amax / torch.tensor(E4M3_MAX), torch.ones_like(amax))
    q = (xf * inv[:, None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q, scale


def gemm(xq: torch.Tensor, xscale: torch.Tensor, wq: torch.Tensor, wscale: torch.Tensor) -> torch.Tensor:
    """fp32 [M][N] = (wscale[n] * xscale[m]) * sum_k xq[m][k] wq[n][k]; float64 accumulation stands in for "exact"."""
    acc = (xq.to(torch.float64) @ wq.to(torch.float64).t()).float()
    return acc * (wscale[None, :] * xscale[:, None])
