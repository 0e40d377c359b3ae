"""The E4M3 KV cache of 256-wide heads (hwocr_kv.fp8: BASELINE config 4's decode attention streams half the bytes) on the MI355X,
operator level, through the C ABI.  The contract is this repo's own (HF has no fp8 path: the fp8 leg is a stated tolerance) and is
pinned like the fp8 GEMM: the quantiser bit for bit against oracle/fp8_ref.py (itself pinned to the OCP encodings), the layout
against its Python restatement, the attention against the fp32 product of the SAME codes and scales.
  * hwocr_kv_quant_fp8          the prefill's cache fill: codes + one scale per token and kv head
  * hwocr_attn_decode_qkv_fp8kv the fused decode step: the new token is quantised and appended, keys / values dequantised in registers"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._gpu_util import DEV, assert_close_bf16, lib, p, randbf, randf32, rbf, st  # noqa: E402
from tests.test_ops_gpu import _rope_tables, _sdpa_ref  # noqa: E402

HD = 256


def k_offsets(ctx):
    """byte offset of code (key, d) inside a (read, kv head) region: csrc/common.h kv8_k restated."""
    key = torch.arange(ctx).view(-1, 1)
    d = torch.arange(HD).view(1, -1)
    kl = key & 31
    t = (kl >> 2) & 1
    c = ((kl >> 3) << 2) | (kl & 3)
    s, qd = d >> 5, (d >> 3) & 3
    return (((((key >> 5) * 2 + t) * 4 + (s >> 1)) * 64 + qd * 16 + c) * 16 + (s & 1) * 8 + (d & 7)).reshape(-1)


def v_offsets(ctx):
    """... of code (d, key): kv8_v."""
    d = torch.arange(HD).view(-1, 1)
    key = torch.arange(ctx).view(1, -1)
    kl = key & 31
    return ((((key >> 5) * 8 + (d >> 5)) * 64 + (kl >> 3) * 16 + (d & 15)) * 16 + ((d >> 4) & 1) * 8 + (kl & 7)).reshape(-1)


def test_layout_maps_are_permutations():
    for ctx in (32, 96, 512):
        for off in (k_offsets(ctx), v_offsets(ctx)):
            assert sorted(off.tolist()) == list(range(ctx * HD))


def _quant(x):
    """oracle quantiser on rows of x [..., 256] -> (codes uint8, scales fp32)."""
    from oracle import fp8_ref

    q, s = fp8_ref.quant_rows(x.reshape(-1, x.shape[-1]).cpu())
    return q.view(torch.uint8).reshape(x.shape), s.reshape(x.shape[:-1])


@pytest.mark.parametrize("nseq,Hkv,keys,ctx", [(3, 1, 64, 128), (2, 2, 4160, 4224)])
def test_kv_quant_fp8_codes_scales_and_layout(nseq, Hkv, keys, ctx):
    k = randbf(nseq, Hkv, keys, HD, seed=41)
    v = randbf(nseq, Hkv, keys, HD, scale=3.0, seed=42)
    k[0, 0, 5] = 0                                                     # an all-zero token: scale 1, codes 0
    vt = v.transpose(2, 3).contiguous()
    K8 = torch.full((nseq, Hkv, ctx * HD), 0xEE, dtype=torch.uint8, device=DEV)
    V8 = torch.full((nseq, Hkv, ctx * HD), 0xEE, dtype=torch.uint8, device=DEV)
    ks = torch.full((nseq, Hkv, ctx), -1.0, dtype=torch.float32, device=DEV)
    vs = torch.full((nseq, Hkv, ctx), -1.0, dtype=torch.float32, device=DEV)
    assert lib().hwocr_kv_quant_fp8(p(k), p(vt), Hkv * keys * HD, keys * HD, Hkv * HD * keys, HD * keys, keys, p(K8), p(V8), p(ks), p(vs),
                                    nseq, Hkv, keys, ctx, st()) == 0
    torch.cuda.synchronize()
    kq, ksc = _quant(k)
    vq, vsc = _quant(v)
    assert torch.equal(ks[:, :, :keys].cpu(), ksc) and torch.equal(vs[:, :, :keys].cpu(), vsc)
    assert bool((ks[:, :, keys:] == -1).all()) and float(ks[0, 0, 5]) == 1.0
    ko, vo = k_offsets(ctx).view(ctx, HD)[:keys].reshape(-1), v_offsets(ctx).view(HD, ctx)[:, :keys].reshape(-1)
    assert torch.equal(K8.cpu()[:, :, ko].view(nseq, Hkv, keys, HD), kq)
    assert torch.equal(V8.cpu()[:, :, vo].view(nseq, Hkv, HD, keys), vq.transpose(2, 3))
    untouched = torch.ones(ctx * HD, dtype=torch.bool)
    untouched[ko] = False
    assert bool((K8.cpu()[:, :, untouched] == 0xEE).all())


@pytest.mark.parametrize("Hq,Hkv,nsplit,B", [(8, 1, 1, 6), (8, 1, 4, 5), (4, 2, 3, 4), (8, 1, 1, 252)])
def test_attn_decode_qkv_over_the_e4m3_cache(Hq, Hkv, nsplit, B):
    """The fused decode step over the E4M3 cache.  (1) The appended token: exactly the oracle quantisation of the bf16 key / value the
    bf16-cache kernel appends (bias, rotary and rounding chain unchanged).  (2) The output: the fp32 attention over the DEQUANTISED
    cache (codes x scales, the new token's included), 4 bf16 ulps as test_attn_decode.  Reads whose new slot opens a block, sits last in
    a block, at the first and the last cache position; with splits: merge launch and last-workgroup merge, identical bytes."""
    ctx, nslab, max_pos = 512, 2, 1024
    W = (Hq + 2 * Hkv) * HD
    G = Hq // Hkv
    g = torch.Generator().manual_seed(11)
    lens = torch.randint(2, ctx, (B,), generator=g).tolist()
    lens[0], lens[1], lens[2], lens[3] = 1, ctx, 33, 64
    delta = torch.randint(-1, 300, (B,), generator=g).tolist()
    delta[0] = 0
    slabs = randf32(nslab, B, W, seed=901)
    cos_t, sin_t = _rope_tables(max_pos, hd=HD)
    cos_d, sin_d = cos_t.to(DEV), sin_t.to(DEV)
    k = randbf(B, Hkv, ctx, HD, seed=16)
    v = randbf(B, Hkv, ctx, HD, scale=2.0, seed=17)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    delta_d = torch.tensor(delta, dtype=torch.int32, device=DEV)
    # what the bf16-cache kernel appends (the reference for the new token's k / v)
    Kb, Vb = k.clone(), v.transpose(2, 3).contiguous()
    outb = torch.zeros(B, Hq * HD, dtype=torch.bfloat16, device=DEV)
    stb = torch.zeros(1, dtype=torch.int32, device=DEV)
    po = torch.zeros(B * Hkv * nsplit * G * HD, dtype=torch.float32, device=DEV)
    pm = torch.zeros(B * Hkv * nsplit * G * 2, dtype=torch.float32, device=DEV)
    assert lib().hwocr_attn_decode_qkv(p(slabs), nslab, B * W, None, p(Kb), p(Vb), p(lens_d), p(delta_d), p(cos_d), p(sin_d), p(outb), p(po),
                                       p(pm), None, B, Hq, Hkv, nsplit, Hkv * ctx * HD, ctx * HD, Hkv * HD * ctx, HD * ctx, ctx, HD ** -0.5,
                                       HD, 0, ctx, max_pos, p(stb), st()) == 0
    torch.cuda.synchronize()
    new_k = torch.stack([Kb[b, :, lens[b] - 1] for b in range(B)])            # [B, Hkv, 256]
    new_v = torch.stack([Vb[b, :, :, lens[b] - 1] for b in range(B)])
    # the E4M3 cache of the OLD tokens, built with the oracle quantiser and the Python layout maps
    kq, ksc = _quant(k)
    vq, vsc = _quant(v)
    ko, vo = k_offsets(ctx), v_offsets(ctx)
    K8 = torch.zeros(B, Hkv, ctx * HD, dtype=torch.uint8)
    V8 = torch.zeros(B, Hkv, ctx * HD, dtype=torch.uint8)
    K8[:, :, ko] = kq.reshape(B, Hkv, ctx * HD)
    V8[:, :, vo] = vq.transpose(2, 3).reshape(B, Hkv, HD * ctx)
    outs = []
    for lastwg in ([False, True] if nsplit > 1 else [False]):
        K8d, V8d, ksd, vsd = K8.to(DEV), V8.to(DEV), ksc.to(DEV).contiguous(), vsc.to(DEV).contiguous()
        arrive = torch.zeros(B * Hkv, dtype=torch.int32, device=DEV) if lastwg else None
        out = torch.zeros(B, Hq * HD, dtype=torch.bfloat16, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        po.zero_()
        pm.zero_()
        assert lib().hwocr_attn_decode_qkv_fp8kv(p(slabs), nslab, B * W, None, p(K8d), p(V8d), p(ksd), p(vsd), p(lens_d), p(delta_d), p(cos_d),
                                                 p(sin_d), p(out), p(po), p(pm), p(arrive), B, Hq, Hkv, nsplit, HD ** -0.5, ctx, max_pos,
                                                 p(status), st()) == 0
        torch.cuda.synchronize()
        assert int(status) == 0 and (arrive is None or int(arrive.abs().sum()) == 0)
        outs.append(out)
        # (1) the appended token
        nkq, nks = _quant(new_k)
        nvq, nvs = _quant(new_v)
        Kc, Vc = K8d.cpu(), V8d.cpu()
        for b in range(B):
            slot = lens[b] - 1
            assert torch.equal(Kc[b][:, ko.view(ctx, HD)[slot]], nkq[b]), f"read {b}: appended key codes"
            assert torch.equal(Vc[b][:, v_offsets(ctx).view(HD, ctx)[:, slot]], nvq[b]), f"read {b}: appended value codes"
            assert torch.equal(ksd[b, :, slot].cpu(), nks[b]) and torch.equal(vsd[b, :, slot].cpu(), nvs[b])
            # nothing else of the cache moved
            keep = torch.ones(ctx * HD, dtype=torch.bool)
            keep[ko.view(ctx, HD)[slot]] = False
            assert torch.equal(Kc[b][:, keep], K8[b][:, keep])
        # (2) the attention over the dequantised cache
        kd = Kc[:, :, ko].view(B, Hkv, ctx, HD).view(torch.float8_e4m3fn).float() * ksd.cpu()[..., None]
        vd = Vc[:, :, vo].view(B, Hkv, HD, ctx).view(torch.float8_e4m3fn).float().transpose(2, 3) * vsd.cpu()[..., None]
        # queries: bf16(sum of slabs), rotated like the key (taken from the bf16 kernel's own arithmetic through its output is not
        # possible; restate: the same chain as tests' decode_qkv_finish reference)
        y = rbf(slabs.sum(0).cpu())                                                       # [B, W]
        for b in (0, 1, 2, 3, B - 1):
            n, pos = lens[b], lens[b] - 1 + delta[b]
            q = y[b, : Hq * HD].view(Hq, HD)
            cs, sn = cos_t[pos].float(), sin_t[pos].float()
            x1, x2 = q[:, : HD // 2], q[:, HD // 2:]
            qr = torch.cat([rbf(rbf(x1 * cs) + rbf(-x2 * sn)), rbf(rbf(x2 * cs) + rbf(x1 * sn))], dim=1)
            want = _sdpa_ref(qr.view(Hq, 1, HD), kd[b, :, :n], vd[b, :, :n], False, HD ** -0.5).reshape(Hq * HD)
            assert_close_bf16(out[b].cpu(), want, ulps=4.0, atol=4e-3, what=f"e4m3-cache attention, read {b} len {n}")
    if len(outs) == 2:
        assert torch.equal(outs[0], outs[1]), "last-workgroup merge differs from the merge launch"
