"""oracle/sampling.py ("hwocr sampling v1", the exact procedure the HIP draw follows) against what it restates: Random123's
Philox4x32-10 known answers, and the kept sets / probabilities of transformers' own Temperature / TopK / TopP warpers (HF
generation/logits_process.py, reached from /root/reference/ocr_agent/tools.py:765 when generation_config.json says do_sample)."""
import numpy as np
import pytest
import torch

from oracle import sampling as S


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert S.philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert S.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert S.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_weights_track_exp2():
    d = np.concatenate([np.linspace(0, 30, 4001), [np.inf, 0.0, 1e-6]]).astype(np.float32)
    for t in (0.3, 1.0, 2.5):
        c = np.float32(np.float32(1.4426950408889634) / np.float32(t))
        w = S.weights(d, c).astype(np.float64) / 2.0 ** 32
        want = np.exp2(-(d.astype(np.float64) * float(c)))
        live = np.isfinite(d)
        assert (np.abs(w[live] - want[live]) <= 2.0 ** -32 + 4e-7 * want[live]).all()  # polynomial error + the fixed-point floor
        assert w[-3] == 0 and w[-2] == 1.0


@pytest.mark.parametrize("seed", range(8))
def test_kept_sets_and_probabilities_equal_the_hf_warpers(seed):
    """Tie-free fp32 scores: the kept set is exactly the warpers', the probabilities agree to 2e-6.  (On ties at a threshold HF
    keeps all of them for top-k, as here, and cuts inside the tie group in sort order for top-p, where this procedure keeps the
    group whole - stated in DESIGN.md.)"""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

    rng = np.random.default_rng(seed)
    V = 8192
    for _ in range(6):
        s = (rng.standard_normal(V) * rng.uniform(1, 4)).astype(np.float32)
        t = float(rng.choice([0.3, 0.7, 1.0, 1.5]))
        k = int(rng.choice([0, 5, 50, 1000]))
        p = float(rng.choice([1.0, 0.95, 0.8, 0.3]))
        keep, w, _ = S.kept_and_weights(s, t, k, p)
        x = TemperatureLogitsWarper(t)(None, torch.tensor(s)[None])
        if k:
            x = TopKLogitsWarper(k)(None, x)
        if p < 1:
            x = TopPLogitsWarper(p)(None, x)
        hf_keep = torch.isfinite(x[0]).numpy()
        hf_p = torch.softmax(x[0].double(), -1).numpy()
        assert (keep == hf_keep).all(), (t, k, p, int(keep.sum()), int(hf_keep.sum()))
        assert np.abs(w / w.sum() - hf_p).max() < 2e-6


def test_draws_follow_the_distribution():
    rng = np.random.default_rng(3)
    V = 64
    s = (rng.standard_normal(V) * 2).astype(np.float32)
    _, w, _ = S.kept_and_weights(s, 0.9, 20, 0.95)
    p = w / w.sum()
    n = 12000
    cnt = np.zeros(V)
    for i in range(n):
        cnt[S.sample(s, 0.9, 20, 0.95, seed=1234, read=i % 7, step=i // 7)[0]] += 1
    live = p > 0
    assert cnt[~live].sum() == 0
    chi2 = ((cnt[live] - n * p[live]) ** 2 / (n * p[live])).sum()
    assert chi2 < 45, chi2  # dof <= 19: p < 1e-3 at 45
    # same (seed, read, step) -> same token; another seed -> another stream
    a = [S.sample(s, 0.9, 20, 0.95, 1, 3, t)[0] for t in range(64)]
    assert a == [S.sample(s, 0.9, 20, 0.95, 1, 3, t)[0] for t in range(64)]
    assert a != [S.sample(s, 0.9, 20, 0.95, 2, 3, t)[0] for t in range(64)]
