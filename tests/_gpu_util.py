"""Helpers for the -m gpu parity tests: every call goes through the C ABI of libhwocr_hip.so."""
import torch

from handwritten_ocr_amd import _lib

DEV = "cuda"


def lib():
    return _lib.hip()


def p(t):
    return _lib.ptr(t)


def st():
    return _lib.stream_handle()


def rbf(x: torch.Tensor) -> torch.Tensor:
    """fp32 value after a round trip through bf16 (a materialised bf16 tensor in the reference's modules)."""
    return x.to(torch.bfloat16).to(torch.float32)


def randbf(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(0 if seed is None else seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def assert_close_bf16(got: torch.Tensor, want_f32: torch.Tensor, ulps: float = 2.0, atol: float = 1e-3, what="",
                      mag: torch.Tensor | None = None):
    """got: bf16 kernel output; want: fp32 reference.  1 bf16 ulp of x = 2^(floor(log2|x|) - 7).
    `mag` (optional) is the magnitude the ulp is taken of when a larger rounded intermediate feeds the output."""
    g = got.float()
    err = (g - want_f32).abs()
    ref_mag = want_f32.abs() if mag is None else torch.maximum(want_f32.abs(), mag)
    ulp = torch.exp2(torch.floor(torch.log2(ref_mag.clamp_min(1e-30))) - 7.0)
    tol = atol + ulps * ulp
    bad = err > tol
    assert not bad.any(), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max err {float(err.max()):.4g} "
        f"at ref {float(want_f32.flatten()[err.argmax()]):.4g}")
