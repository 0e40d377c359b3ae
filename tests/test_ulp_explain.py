"""The explained-outlier rule of the GEMM epilogue parity tests (tests/_gpu_util.assert_close_bf16_explained), checked on the CPU:
an output that is one rounding flip of an intermediate away from the reference is accepted, anything else beyond the hard bound is
not — so the rule cannot pass an indexing slip as a "coincidence" (VERDICT r3, weak #2)."""
import pytest
import torch

from tests._gpu_util import assert_close_bf16_explained, bf16_neighbours, rbf
from tests.test_ops_gpu import _activation_candidates, _epilogue_ref, _gated_candidates


def test_bf16_neighbours_are_one_ulp_apart():
    x = rbf(torch.tensor([1.0, -3.5, 0.0078125, 255.0]))
    n = bf16_neighbours(x)
    assert torch.equal(n[:, 1], x)
    assert torch.equal(n[:, 2] - n[:, 1], torch.tensor([2 ** -7, -2 ** -6, 2 ** -14, 1.0]))   # (the bit pattern's order: away from zero)
    assert torch.equal(n.to(torch.bfloat16).float(), n)


@pytest.mark.parametrize("epi", [2, 3, 6])
def test_flips_of_the_rounded_intermediates_are_explained_and_garbage_is_not(epi):
    g = torch.Generator().manual_seed(epi)
    v = torch.randn(4096, generator=g) * 2.0
    bias = torch.zeros(4096)
    want = _epilogue_ref(v.view(1, -1), bias, None, epi).flatten()
    out = want.to(torch.bfloat16)
    # element 7: x = bf16(acc + bias) landed one ulp up AND (quick-GELU) the rounded gate one ulp down: up to ~2.5 ulps away
    x_up = bf16_neighbours(rbf(v[7:8]))[:, 2]
    if epi == 2:
        s_dn = bf16_neighbours(rbf(torch.sigmoid(rbf(1.702 * x_up))))[:, 0]
        flipped = x_up * s_dn
    else:
        flipped = _epilogue_ref(x_up.view(1, 1), torch.zeros(1), None, epi).flatten()
    out[7] = flipped.to(torch.bfloat16)[0]
    cands = lambda idx: _activation_candidates(v[idx], epi)  # noqa: E731
    # hard bound 0 ulps here: every element that differs from the reference's rounding at all must be explained
    n = assert_close_bf16_explained(out.view(1, -1), want.view(1, -1), ulps=0.51, atol=0.0, what="flip", mag=None, candidates=cands,
                                    max_frac=1e-3)
    assert n <= 1
    out[11] = (want[11] * 1.05 + 0.03).to(torch.bfloat16)          # a wrong value: several ulps away, no flip produces it
    with pytest.raises(AssertionError, match="NOT one rounding flip away"):
        assert_close_bf16_explained(out.view(1, -1), want.view(1, -1), ulps=2.0, atol=0.0, what="garbage", mag=None,
                                    candidates=cands, max_frac=1e-3)
    out[11] = want[11].to(torch.bfloat16)
    # many outliers, each of them a genuine flip of ITS OWN intermediate: explained one by one, but no longer rare
    xs_up = bf16_neighbours(rbf(v[100:200]))[:, 2]
    out[100:200] = _epilogue_ref(xs_up.view(1, -1), torch.zeros(100), None, epi).flatten().to(torch.bfloat16)
    with pytest.raises(AssertionError, match="not rare coincidences"):
        assert_close_bf16_explained(out.view(1, -1), want.view(1, -1), ulps=0.51, atol=0.0, what="many", mag=None, candidates=cands)


def test_gated_candidates_cover_a_gate_and_an_up_flip():
    g_pre, u_pre = torch.tensor([-2.37, 0.81]), torch.tensor([1.93, -0.44])
    c = _gated_candidates(g_pre, u_pre, geglu=False)
    assert c.shape == (2, 27)
    g_dn = bf16_neighbours(rbf(g_pre))[:, 0]
    u_up = bf16_neighbours(rbf(u_pre))[:, 2]
    v = rbf(torch.nn.functional.silu(g_dn)) * u_up
    assert bool(((c - v.unsqueeze(1)).abs() < 1e-7).any(dim=1).all())
