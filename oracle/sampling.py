"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

The token draw of `model.generate(..., do_sample=True)` — what the reference's run_ocr gets when the checkpoint's
generation_config.json switches sampling on (/root/reference/ocr_agent/tools.py:765 passes only max_new_tokens) — restated as an
exact integer procedure, "hwocr sampling v1".  HF's pipeline (transformers generation/logits_process.py: RepetitionPenalty ->
Temperature -> TopK -> TopP warpers; generation/utils.py `_sample`: softmax + torch.multinomial) fixes the DISTRIBUTION; its draw
comes from torch's global Philox stream, which no other implementation reproduces token for token.  This procedure keeps HF's
distribution (kept sets as the warpers define them, up to the tie rules below) and pins the draw to a counter-based RNG so that a
read's tokens depend only on (seed, read index, step) — not on batch layout, graph replay or device:

  s_i   fp32 score after repetition penalty / EOS suppression (as the greedy path computes it); M = max s_i; d_i = fp32(M - s_i)
  q_i   selection key = min(2^22 - 1, floor(d_i * 65536))   (distance from the maximum in 2^-16 steps, clamped at 64)
  top_k (0 < k < V): t = k-th smallest q; keep q_i <= t      (HF: scores < k-th largest removed; ties and keys equal at 2^-16 kept)
  w_i   weight = floor(2^32 * exp2(-d_i * c)), c = fp32(log2(e) / temperature); exp2 by the degree-6 polynomial below in NON-fused
        fp32 arithmetic (bit-reproducible on any IEEE machine); removed tokens have w_i = 0
  top_p (< 1): W = sum w_i; P = max(1, floor(float64(top_p) * float64(W))); tau = the smallest key whose inclusive mass from the
        top reaches P; keep q_i <= tau     (HF: ascending cumulative probability <= 1 - top_p removed; min_tokens_to_keep = 1)
  draw  r = Philox4x32-10(counter = (read, step, 0, 0), key = (seed_lo, seed_hi)); r64 = r[1] << 32 | r[0];
        target = (W_kept * r64) >> 64; token = the smallest id whose running kept mass (in id order) exceeds target

Everything after w_i is integer arithmetic: the HIP kernel (csrc/elementwise.hip: sample_advance_kernel) must agree bit for bit.

Known deviation from HF's TopKLogitsWarper (the kernel mirrors it bit for bit, so both deviate): keys are clamped at 64
logits below the maximum, and a tie group at the threshold is kept whole.  If FEWER than k tokens lie within 64 logits of the
maximum, the k-th key is the clamp itself and every token passes `q <= t`: the tokens HF would have cut keep their weight
floor(2^32 * 2^(-d log2e / T)).  At d >= 64 that weight is 0 for T <= 2.88 (2^32 * 2^(-64 * 1.4427 / T) < 1), so the kept SET differs
from HF's but the distribution does not; above that temperature each such token carries a weight of a few units in 2^32 (483 at
T = 4, i.e. 1.1e-7 of the maximum's; V = 152 k tokens all sitting exactly 64 below the maximum: 1.7e-2 of its mass, the worst case).  The
checkpoints of this path ship T <= 1; tests/test_sampling_oracle.py compares with HF's warpers in the un-clamped regime only."""
from __future__ import annotations

import numpy as np

QMAX = (1 << 22) - 1
_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
# 2^f on [-0.5, 0.5]: Taylor coefficients of exp(f ln 2) to degree 6 (error < 1.3e-7), fp32
_C = [np.float32(x) for x in (1.0, 0.6931471805599453, 0.2402265069591007, 0.05550410866482158, 0.009618129107628477,
                              0.0013333558146428443, 0.00015403530393381608)]


def philox4x32_10(counter, key):
    """Random123 Philox4x32-10.  counter: 4 uint32, key: 2 uint32 -> 4 uint32."""
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + _W0) & 0xFFFFFFFF, (k[1] + _W1) & 0xFFFFFFFF]
    return c


def weights(d: np.ndarray, c: np.float32) -> np.ndarray:
    """w = floor(2^32 * 2^(-d c)) for fp32 distances d >= 0 (inf allowed), as uint64, in non-fused fp32 arithmetic."""
    d = d.astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        y = -(d * np.float32(c))            # fp32 product, then negation (exact)
        dead = ~(y > np.float32(-40.0))     # also catches -inf and nan
        y = np.where(dead, np.float32(0), y).astype(np.float32)
        n = np.rint(y).astype(np.float32)   # round half to even (v_rndne_f32)
        f = (y - n).astype(np.float32)
        r = _C[6]
        for k in range(5, -1, -1):
            r = (r * f).astype(np.float32)
            r = (r + _C[k]).astype(np.float32)
        w = np.floor(r.astype(np.float64) * np.exp2(n.astype(np.float64) + 32.0)).astype(np.uint64)
    return np.where(dead, np.uint64(0), w)


def keys(s: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(q, d) of fp32 scores s (may hold -inf): distance keys and the distances they come from."""
    s = s.astype(np.float32)
    m = np.float32(s.max())
    with np.errstate(invalid="ignore", over="ignore"):
        d = (m - s).astype(np.float32)
        t = np.floor((d * np.float32(65536.0)).astype(np.float32))
    q = np.where(t >= np.float32(QMAX), QMAX, np.where(np.isfinite(t), t, QMAX)).astype(np.int64)
    return q, d


def kept_and_weights(s: np.ndarray, temperature: float, top_k: int, top_p: float):
    """-> (kept mask, uint64 weights of the kept tokens (0 elsewhere), debug dict)."""
    V = s.shape[0]
    q, d = keys(s)
    c = np.float32(np.float32(1.4426950408889634) / np.float32(temperature))
    keep = np.ones(V, bool)
    dbg = {}
    if 0 < top_k < V:
        t = int(np.partition(q, top_k - 1)[top_k - 1])
        keep &= q <= t
        dbg["tk"] = t
    w = np.where(keep, weights(d, c), np.uint64(0)).astype(np.uint64)
    W = int(w.sum(dtype=np.uint64))
    dbg["W"] = W
    if top_p < 1.0:
        P = max(1, int(np.floor(np.float64(np.float32(top_p)) * np.float64(W))))
        order = np.argsort(q, kind="stable")
        cum = np.cumsum(w[order], dtype=np.uint64)  # exact: V * 1.5 * 2^32 < 2^64
        j = min(int(np.searchsorted(cum, np.uint64(P), side="left")), V - 1)
        tau = int(q[order][j])
        keep &= q <= tau
        w = np.where(keep, w, np.uint64(0)).astype(np.uint64)
        dbg["P"], dbg["tau"] = P, tau
    dbg["Wk"] = int(w.sum(dtype=np.uint64))
    return keep, w, dbg


def sample(s: np.ndarray, temperature: float, top_k: int, top_p: float, seed: int, read: int, step: int):
    """One draw: (token id, debug dict)."""
    keep, w, dbg = kept_and_weights(s, temperature, top_k, top_p)
    r = philox4x32_10((read, step, 0, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    r64 = (r[1] << 32) | r[0]
    target = (dbg["Wk"] * r64) >> 64
    cum = 0
    tok = int(np.flatnonzero(keep)[-1]) if keep.any() else 0
    for i in np.flatnonzero(w):
        cum += int(w[i])
        if cum > target:
            tok = int(i)
            break
    dbg["target"] = target
    return tok, dbg


def penalized_scores(logits_bf16_as_f32: np.ndarray, seen: np.ndarray | None, rep_penalty: float, suppress_eos=()):
    """The fp32 scores the selection kernels start from: repetition penalty on ids in `seen` (bool mask), EOS -> -inf."""
    s = logits_bf16_as_f32.astype(np.float32).copy()
    if seen is not None and rep_penalty != 1.0:
        p = np.float32(rep_penalty)
        s = np.where(seen, np.where(s < 0, (s * p).astype(np.float32), (s / p).astype(np.float32)), s).astype(np.float32)
    for e in suppress_eos:
        s[e] = -np.inf
    return s
