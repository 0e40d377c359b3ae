"""The fp8 oracle against the OCP FP8 (E4M3) encodings and its own quantisation contract (oracle/fp8_ref.py)."""
import torch

from oracle import fp8_ref


def _bits(v: float) -> int:
    return int(torch.tensor([v]).to(torch.float8_e4m3fn).view(torch.uint8)[0])


def test_e4m3_encodings_of_the_ocp_spec():
    # sign 1, exponent 4 (bias 7), mantissa 3; S.1111.111 is NaN, so the largest finite value is S.1111.110 = 448
    assert _bits(448.0) == 0x7E and _bits(-448.0) == 0xFE
    assert _bits(1.0) == 0x38 and _bits(0.5) == 0x30 and _bits(2.0) == 0x40
    assert _bits(2.0 ** -6) == 0x08            # smallest normal
    assert _bits(2.0 ** -9) == 0x01            # smallest subnormal
    assert _bits(0.0) == 0x00 and _bits(-0.0) == 0x80
    # round-to-nearest-even on ties: 1.0625 is halfway 1.0 (mantissa 000) .. 1.125 (001); 1.1875 halfway 1.125 .. 1.25 (010)
    assert _bits(1.0625) == 0x38 and _bits(1.1875) == 0x3A
    # 464 is halfway 448 .. (non-existent) 480: values the quantiser can produce never exceed 448 by more than rounding
    assert _bits(447.9) == 0x7E


def test_quant_rows_contract():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(5, 256, generator=g) * 3).to(torch.bfloat16)
    x[2] = 0
    x[3, 7] = 1000.0
    q, s = fp8_ref.quant_rows(x)
    assert q.dtype == torch.float8_e4m3fn and s.dtype == torch.float32
    assert float(s[2]) == 1.0 and not q[2].float().any()
    assert float(q[3, 7].float()) == 448.0 and abs(float(s[3]) - 1000.0 / 448.0) < 1e-6
    # the row maximum always lands on +-448; dequantised values are within half an fp8 step (2^-4 relative) of the input
    assert torch.equal(q.float().abs().amax(-1)[[0, 1, 3, 4]], torch.full((4,), 448.0))
    back = q.float() * s[:, None]
    assert float(((back - x.float()).abs() / x.float().abs().clamp_min(float(s.max()) * 2 ** -6)).max()) <= 2 ** -4 + 1e-6


def test_gemm_is_the_scaled_integer_product():
    xq = torch.tensor([[1.0, 2.0, -4.0, 0.5]]).to(torch.float8_e4m3fn)
    wq = torch.tensor([[2.0, 1.0, 1.0, 8.0], [0.0, -1.0, 0.25, 0.0]]).to(torch.float8_e4m3fn)
    out = fp8_ref.gemm(xq, torch.tensor([0.5]), wq, torch.tensor([2.0, 4.0]))
    assert out.tolist() == [[4.0, -6.0]]
