"""hwocr_sample_advance (generate(do_sample=True): temperature -> top-k -> top-p -> draw) against oracle/sampling.py, which is exact
integer arithmetic: thresholds, masses, draw target and token must agree bit for bit - on bf16 logits with their many exact ties, with
repetition penalty and EOS suppression, at the real vocabulary widths.  Plus the distribution of the device's draws (chi-square) and
the engine path (tests/test_model_gpu.py: test_sampled_reads_depend_on_seed_read_and_step_only)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampling as S  # noqa: E402
from tests._gpu_util import DEV, lib, p, st  # noqa: E402


def sync():
    torch.cuda.synchronize()


def _run(logits, temperature, top_k, top_p, seed, read_ids, n_gen, seen=None, rep=1.0, min_new=0, eos=(), max_new=8):
    R, V = logits.shape
    i32 = dict(dtype=torch.int32, device=DEV)
    cur, lens, fin = torch.zeros(R, **i32), torch.ones(R, **i32), torch.zeros(R, **i32)
    ng = torch.tensor(n_gen, **i32)
    out = torch.full((R, max_new), -7, **i32)
    dbg = torch.zeros(R, 8, dtype=torch.int64, device=DEV)
    rid = torch.tensor(read_ids, **i32)
    eos_arr = (C.c_int * 4)(*(list(eos) + [0] * 4)[:4])
    seen_ld = (V + 31) // 32
    seen_t = None
    if seen is not None:
        words = np.packbits(seen.astype(np.uint8), axis=1, bitorder="little")
        words = np.pad(words, ((0, 0), (0, seen_ld * 4 - words.shape[1]))).view(np.uint32).astype(np.int64).astype(np.int32, casting="unsafe")
        seen_t = torch.from_numpy(words.reshape(R, seen_ld)).to(DEV)
    rc = lib().hwocr_sample_advance(p(logits), V, V, R, p(cur), p(lens), p(ng), p(fin), p(out), max_new, min_new, eos_arr, len(eos), 0,
                                    p(seen_t) if seen_t is not None else None, seen_ld, rep, temperature, top_k, top_p, seed, p(rid), p(dbg), st())
    assert rc == 0
    sync()
    return cur.cpu().numpy(), dbg.cpu().numpy().astype(np.uint64), out.cpu().numpy(), ng.cpu().numpy(), fin.cpu().numpy()


@pytest.mark.parametrize("V,temperature,top_k,top_p", [
    (4096, 0.7, 50, 0.9), (4096, 1.0, 0, 1.0), (4096, 1.3, 5, 1.0), (4096, 0.5, 0, 0.5), (151936, 0.8, 50, 0.95), (151936, 1.0, 0, 0.9),
    (257216, 0.6, 20, 1.0), (4096, 4.0, 3000, 0.999), (512, 1.0, 511, 0.2)])
def test_draw_equals_the_oracle(V, temperature, top_k, top_p):
    R = 6
    g = torch.Generator(device="cpu").manual_seed(V + top_k)
    logits = (torch.randn(R, V, generator=g) * 3.0).to(torch.bfloat16)
    logits[1, : V // 2] = logits[1, 0]            # a plateau of exact ties
    logits[2] = logits[2] * 0.02                  # nearly flat: thresholds deep in the bulk
    seen = (torch.rand(R, V, generator=g) < 0.02).numpy()
    eos = (3, V - 1)
    n_gen = [0, 1, 2, 5, 0, 3]
    read_ids = [11, 7, 0, 100000, 2, 3]
    lf = logits.float().numpy()
    tok, dbg, out, ng, fin = _run(logits.to(DEV), temperature, top_k, top_p, 0x1234_5678_9ABC_DEF0, read_ids, n_gen, seen=seen, rep=1.3,
                                  min_new=2, eos=eos)
    for r in range(R):
        if n_gen[r] > 0:
            seen[r, 0] = True  # the token the step was fed (cur_ids = 0 here) joins the bitmap before the scores are read
        s = S.penalized_scores(lf[r], seen[r], 1.3, suppress_eos=eos if n_gen[r] < 2 else ())
        want, d = S.sample(s, temperature, top_k, top_p, 0x1234_5678_9ABC_DEF0, read_ids[r], n_gen[r])
        assert int(dbg[r, 0]) == int(np.float32(s.max()).view(np.uint32)), "maximum"
        if 0 < top_k < V:
            assert int(dbg[r, 1]) == d["tk"], ("top-k key", r)
        if top_p < 1:
            assert (int(dbg[r, 2]), int(dbg[r, 3]), int(dbg[r, 4])) == (d["W"], d["P"], d["tau"]), ("nucleus", r)
        assert (int(dbg[r, 5]), int(dbg[r, 6])) == (d["Wk"], d["target"]), ("kept mass / target", r)
        assert tok[r] == want, (r, int(tok[r]), want)
        assert out[r, n_gen[r]] == want and ng[r] == n_gen[r] + 1 and fin[r] == int(want in eos)


def test_draws_follow_the_distribution():
    V, R = 512, 256
    g = torch.Generator(device="cpu").manual_seed(5)
    row = (torch.randn(V, generator=g) * 2.0).to(torch.bfloat16)
    logits = row[None].repeat(R, 1).contiguous().to(DEV)
    _, w, _ = S.kept_and_weights(row.float().numpy(), 0.9, 40, 0.97)
    prob = w / w.sum()
    cnt = np.zeros(V)
    for step in range(40):
        tok, *_ = _run(logits, 0.9, 40, 0.97, 99, list(range(R)), [step] * R, max_new=64)
        np.add.at(cnt, tok, 1)
    n = 40 * R
    live = prob > 0
    assert cnt[~live].sum() == 0
    chi2 = ((cnt[live] - n * prob[live]) ** 2 / (n * prob[live])).sum()
    assert chi2 < 2.2 * live.sum() + 30, (chi2, int(live.sum()))
