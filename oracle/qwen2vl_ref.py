"""ORACLE — test infrastructure, not product code.

CPU restatement (plain torch ops, batch 1, no transformers import) of the arithmetic the reference's
``run_ocr`` executes for the Qwen2-VL and Qwen2.5-VL (olmOCR-2, the reference's default OCR_MODEL, config.py:15)
families: ``model.generate`` at ocr_agent/tools.py:764-765, i.e. the
third-party library transformers (reference pins 5.1.0, poetry.lock:5252-5253; validated here against the
installed 5.15.0).  The algorithm lives in that dependency, so every function cites the HF file:line it follows
(HF = site-packages/transformers).

Pinned by tests/golden/qwen2vl_tiny_*.safetensors and qwen25vl_tiny_*.safetensors: outputs of the real HF classes on seeded random-init models,
written by tools/make_goldens.py in the build container (the reference repo itself holds no test or fixture for
this path — SURVEY.md §4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class RefConfig:
    # vision tower (HF models/qwen2_vl/configuration_qwen2_vl.py:31-41)
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: float = 4.0
    patch_size: int = 14
    merge: int = 2
    tps: int = 2
    # Qwen2.5-VL tower (HF models/qwen2_5_vl/configuration_qwen2_5_vl.py:36-66); family "qwen2_vl" ignores these
    family: str = "qwen2_vl"
    vit_inter: int = 0
    window_size: int = 112
    fullatt: tuple = (7, 15, 23, 31)
    # decoder (… :83-102)
    hidden: int = 1536
    layers: int = 28
    q_heads: int = 12
    kv_heads: int = 2
    inter: int = 8960
    vocab: int = 151936
    rope_theta: float = 1_000_000.0
    mrope_section: tuple = (16, 24, 24)
    eps: float = 1e-6
    image_token_id: int = 151655
    vision_start_id: int = 151652
    vision_end_id: int = 151653
    tie: bool = True
    eos_ids: tuple = (151645, 151643)
    pad_id: int = 151643

    @property
    def head_dim(self) -> int:
        return self.hidden // self.q_heads

    @property
    def vit_head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def mlp_dim(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def patch_k(self) -> int:
        return 3 * self.tps * self.patch_size * self.patch_size


# ---------------------------------------------------------------------------------------------- small ops
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """HF modeling_qwen2_vl.py:105-110: fp32 normalise, cast back, then multiply by the weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """HF modeling_qwen2_vl.py:173-177."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """HF activations.py:117-123."""
    return x * torch.sigmoid(1.702 * x)


def vision_position_ids(gh: int, gw: int, merge: int) -> torch.Tensor:
    """(h, w) index of every patch in merge-block-major order.  HF vision_utils.py:81-127."""
    hp = torch.arange(gh).view(-1, 1).expand(gh, gw)
    wp = torch.arange(gw).view(1, -1).expand(gh, gw)

    def blockify(t):
        return t.reshape(gh // merge, merge, gw // merge, merge).permute(0, 2, 1, 3).reshape(-1)

    return torch.stack([blockify(hp), blockify(wp)], dim=-1)


def window_index(gh: int, gw: int, merge: int, window_size: int, patch: int):
    """Window regrouping of one image's merged tokens.  HF vision_utils.py:130-188: the merged grid is padded with -100
    to whole windows of side window_size // merge // patch (a full extra window when already a multiple), windows are
    flattened row-major and the padding dropped.  Returns (window_index [lh*lw], patches per non-empty window)."""
    side = window_size // merge // patch
    lh, lw = gh // merge, gw // merge
    idx = torch.arange(lh * lw).reshape(lh, lw)
    ph, pw = side - lh % side, side - lw % side
    padded = F.pad(idx, (0, pw, 0, ph), value=-100)
    nh, nw = (lh + ph) // side, (lw + pw) // side
    win = padded.reshape(nh, side, nw, side).permute(0, 2, 1, 3).reshape(nh * nw, side * side)
    counts = (win != -100).sum(-1)
    flat = win.reshape(-1)
    return flat[flat != -100], (counts[counts > 0] * merge * merge).tolist()


def rope_index(input_ids: torch.Tensor, image_token_id: int, grids: list[tuple[int, int, int]], merge: int):
    """3-axis M-RoPE position ids for one sequence.  HF modeling_qwen2_vl.py:944-1058 (get_rope_index) +
    :878-925 (get_vision_position_ids).  Returns (pos [3,T] int64, rope_delta int)."""
    ids = input_ids.tolist()
    pos_chunks = []
    cur = 0
    i = 0
    g = iter(grids)
    T = len(ids)
    while i < T:
        j = i
        is_img = ids[i] == image_token_id
        while j < T and (ids[j] == image_token_id) == is_img:
            j += 1
        if not is_img:
            n = j - i
            pos_chunks.append(torch.arange(n).view(1, -1).expand(3, -1) + cur)
            cur += n
        else:
            t, h, w = next(g)
            lh, lw = h // merge, w // merge
            assert t * lh * lw == j - i, "image placeholder run does not match its grid"
            tt = torch.arange(t).view(-1, 1, 1).expand(t, lh, lw).reshape(-1)
            hh = torch.arange(lh).view(1, -1, 1).expand(t, lh, lw).reshape(-1)
            ww = torch.arange(lw).view(1, 1, -1).expand(t, lh, lw).reshape(-1)
            pos_chunks.append(torch.stack([tt, hh, ww]) + cur)
            cur += max(h, w) // merge
        i = j
    pos = torch.cat(pos_chunks, dim=1)
    return pos, int(pos.max()) + 1 - T


class Qwen2VLRef:
    """Functional restatement over a state dict with HF parameter names (model.visual.*, model.language_model.*)."""

    def __init__(self, cfg: RefConfig, sd: dict):
        self.c = cfg
        self.sd = sd
        self.dtype = sd["model.language_model.embed_tokens.weight"].dtype
        self.trace: dict = {}

    def w(self, name: str) -> torch.Tensor:
        return self.sd[name]

    # ------------------------------------------------------------------------------------------ vision tower
    def vision(self, pixel_values: torch.Tensor, grids: list[tuple[int, int, int]]) -> torch.Tensor:
        """HF modeling_qwen2_vl.py:700-729.  pixel_values fp32 [P, 1176] (all images concatenated)."""
        c = self.c
        if c.family == "qwen2_5_vl":
            return self.vision25(pixel_values, grids)
        P = pixel_values.shape[0]
        pre = "model.visual."
        # PatchEmbed: Conv3d with kernel == stride == whole patch, no bias == one matmul (:266-274)
        wpe = self.w(pre + "patch_embed.proj.weight").reshape(c.embed_dim, -1)
        x = F.linear(pixel_values.to(self.dtype), wpe)
        self.trace["patch_embed"] = x
        # 2-D rotary table, fp32 (:239-248, :670-671; position ids vision_utils.py:81-127)
        hd = c.vit_head_dim
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
        pos = torch.cat([vision_position_ids(h, w, c.merge).repeat(t, 1) for (t, h, w) in grids], dim=0)
        rot = (pos.unsqueeze(-1) * inv).flatten(1)  # [P, hd/2] = [h freqs | w freqs]
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos().unsqueeze(-2), emb.sin().unsqueeze(-2)
        seg = [0]
        for (t, h, w) in grids:
            for _ in range(t):
                seg.append(seg[-1] + h * w)
        for l in range(c.depth):
            b = f"{pre}blocks.{l}."
            y = F.layer_norm(x, (c.embed_dim,), self.w(b + "norm1.weight"), self.w(b + "norm1.bias"), 1e-6)
            qkv = F.linear(y, self.w(b + "attn.qkv.weight"), self.w(b + "attn.qkv.bias"))
            q, k, v = qkv.reshape(P, 3, c.num_heads, hd).permute(1, 0, 2, 3).unbind(0)
            qf, kf = q.float(), k.float()  # rotary in fp32, one rounding (:225-236)
            q = (qf * cos + rotate_half(qf) * sin).to(self.dtype)
            k = (kf * cos + rotate_half(kf) * sin).to(self.dtype)
            outs = []
            for s in range(len(seg) - 1):  # one non-causal segment per image (:394-418)
                sl = slice(seg[s], seg[s + 1])
                o = F.scaled_dot_product_attention(q[sl].transpose(0, 1).unsqueeze(0), k[sl].transpose(0, 1).unsqueeze(0),
                                                   v[sl].transpose(0, 1).unsqueeze(0), scale=hd ** -0.5)
                outs.append(o.squeeze(0).transpose(0, 1).reshape(seg[s + 1] - seg[s], c.embed_dim))
            a = torch.cat(outs, dim=0)
            x = x + F.linear(a, self.w(b + "attn.proj.weight"), self.w(b + "attn.proj.bias"))
            y = F.layer_norm(x, (c.embed_dim,), self.w(b + "norm2.weight"), self.w(b + "norm2.bias"), 1e-6)
            y = quick_gelu(F.linear(y, self.w(b + "mlp.fc1.weight"), self.w(b + "mlp.fc1.bias")))
            x = x + F.linear(y, self.w(b + "mlp.fc2.weight"), self.w(b + "mlp.fc2.bias"))
            if l == 0:
                self.trace["vit_block0"] = x
        self.trace["vit_last"] = x
        # PatchMerger (:277-290)
        m = pre + "merger."
        y = F.layer_norm(x, (c.embed_dim,), self.w(m + "ln_q.weight"), self.w(m + "ln_q.bias"), 1e-6)
        y = y.view(-1, c.embed_dim * c.merge * c.merge)
        y = F.gelu(F.linear(y, self.w(m + "mlp.0.weight"), self.w(m + "mlp.0.bias")))
        y = F.linear(y, self.w(m + "mlp.2.weight"), self.w(m + "mlp.2.bias"))
        self.trace["merger"] = y
        return y

    def vision25(self, pixel_values: torch.Tensor, grids: list[tuple[int, int, int]]) -> torch.Tensor:
        """Qwen2.5-VL tower, HF modeling_qwen2_5_vl.py:407-481: RMSNorm blocks, biased SiLU-gated MLP (:84-96), tokens
        regrouped window by window (:441-450), attention inside windows except in `fullatt` layers (:452-465), merger
        output put back in raster order (:474-476).  Single-frame images only (t == 1), as the reference feeds."""
        c = self.c
        P = pixel_values.shape[0]
        pre = "model.visual."
        mm = c.merge * c.merge
        x = F.linear(pixel_values.to(self.dtype), self.w(pre + "patch_embed.proj.weight").reshape(c.embed_dim, -1))
        self.trace["patch_embed"] = x
        hd = c.vit_head_dim
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
        pos = torch.cat([vision_position_ids(h, w, c.merge) for (_, h, w) in grids], dim=0)
        rot = (pos.unsqueeze(-1) * inv).flatten(1)
        order, win, seg = [], [0], [0]
        base = 0
        for (t, h, w) in grids:
            assert t == 1
            wi, lens = window_index(h, w, c.merge, c.window_size, c.patch_size)
            order.append(wi + base)
            for n in lens:
                win.append(win[-1] + n)
            base += h * w // mm
            seg.append(seg[-1] + h * w)
        order = torch.cat(order)
        x = x.reshape(P // mm, mm, -1)[order].reshape(P, -1)
        rot = rot.reshape(P // mm, mm, -1)[order].reshape(P, -1)
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos().unsqueeze(-2), emb.sin().unsqueeze(-2)
        for l in range(c.depth):
            b = f"{pre}blocks.{l}."
            y = rms_norm(x, self.w(b + "norm1.weight"), 1e-6)
            qkv = F.linear(y, self.w(b + "attn.qkv.weight"), self.w(b + "attn.qkv.bias"))
            q, k, v = qkv.reshape(P, 3, c.num_heads, hd).permute(1, 0, 2, 3).unbind(0)
            qf, kf = q.float(), k.float()
            q = (qf * cos + rotate_half(qf) * sin).to(self.dtype)
            k = (kf * cos + rotate_half(kf) * sin).to(self.dtype)
            cu = seg if l in c.fullatt else win
            outs = []
            for s in range(len(cu) - 1):
                sl = slice(cu[s], cu[s + 1])
                o = F.scaled_dot_product_attention(q[sl].transpose(0, 1).unsqueeze(0), k[sl].transpose(0, 1).unsqueeze(0),
                                                   v[sl].transpose(0, 1).unsqueeze(0), scale=hd ** -0.5)
                outs.append(o.squeeze(0).transpose(0, 1).reshape(cu[s + 1] - cu[s], c.embed_dim))
            a = torch.cat(outs, dim=0)
            x = x + F.linear(a, self.w(b + "attn.proj.weight"), self.w(b + "attn.proj.bias"))
            y = rms_norm(x, self.w(b + "norm2.weight"), 1e-6)
            g = F.silu(F.linear(y, self.w(b + "mlp.gate_proj.weight"), self.w(b + "mlp.gate_proj.bias")))
            u = F.linear(y, self.w(b + "mlp.up_proj.weight"), self.w(b + "mlp.up_proj.bias"))
            x = x + F.linear(g * u, self.w(b + "mlp.down_proj.weight"), self.w(b + "mlp.down_proj.bias"))
            if l == 0:
                self.trace["vit_block0"] = x
        self.trace["vit_last"] = x
        m = pre + "merger."
        y = rms_norm(x, self.w(m + "ln_q.weight"), 1e-6).view(-1, c.embed_dim * mm)
        y = F.gelu(F.linear(y, self.w(m + "mlp.0.weight"), self.w(m + "mlp.0.bias")))
        y = F.linear(y, self.w(m + "mlp.2.weight"), self.w(m + "mlp.2.bias"))
        y = y[torch.argsort(order)]
        self.trace["merger"] = y
        return y

    # ------------------------------------------------------------------------------------------ decoder
    def _rope_cos_sin(self, pos3: torch.Tensor):
        """HF modeling_qwen2_vl.py:157-170 + :196-222: fp32 angles, cast to model dtype, sections interleaved."""
        c = self.c
        hd = c.head_dim
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        fr = pos3.float().unsqueeze(-1) * inv  # [3, T, hd/2]
        emb = torch.cat((fr, fr), dim=-1)
        cos, sin = emb.cos().to(self.dtype), emb.sin().to(self.dtype)
        sec = list(c.mrope_section) * 2
        pick = lambda t: torch.cat([ch[i % 3] for i, ch in enumerate(t.split(sec, dim=-1))], dim=-1)
        return pick(cos), pick(sin)  # [T, hd]

    def decoder(self, h: torch.Tensor, pos3: torch.Tensor, cache: list | None):
        """Qwen2VLTextModel layers + final norm (HF modeling_qwen2_vl.py:790-872); h [T, hidden]; cache = per-layer
        (K [kvh, ctx, hd], V [kvh, ctx, hd]) appended in place of HF's DynamicCache (cache_utils.py:127-145)."""
        c = self.c
        T = h.shape[0]
        hd = c.head_dim
        cos, sin = self._rope_cos_sin(pos3)
        cos, sin = cos.unsqueeze(0), sin.unsqueeze(0)
        for l in range(c.layers):
            p = f"model.language_model.layers.{l}."
            x = rms_norm(h, self.w(p + "input_layernorm.weight"), c.eps)
            q = F.linear(x, self.w(p + "self_attn.q_proj.weight"), self.w(p + "self_attn.q_proj.bias"))
            k = F.linear(x, self.w(p + "self_attn.k_proj.weight"), self.w(p + "self_attn.k_proj.bias"))
            v = F.linear(x, self.w(p + "self_attn.v_proj.weight"), self.w(p + "self_attn.v_proj.bias"))
            q = q.view(T, c.q_heads, hd).transpose(0, 1)
            k = k.view(T, c.kv_heads, hd).transpose(0, 1)
            v = v.view(T, c.kv_heads, hd).transpose(0, 1)
            q = (q * cos) + (rotate_half(q) * sin)  # model dtype: every product and the sum are rounded
            k = (k * cos) + (rotate_half(k) * sin)
            past = 0
            if cache is not None:
                if cache[l] is not None:
                    past = cache[l][0].shape[1]
                    k = torch.cat([cache[l][0], k], dim=1)
                    v = torch.cat([cache[l][1], v], dim=1)
                cache[l] = (k, v)
            g = c.q_heads // c.kv_heads
            kk = k.repeat_interleave(g, dim=0)
            vv = v.repeat_interleave(g, dim=0)
            mask = None
            if T > 1:
                mask = torch.ones(T, past + T, dtype=torch.bool).tril(past)
            a = F.scaled_dot_product_attention(q.unsqueeze(0), kk.unsqueeze(0), vv.unsqueeze(0), attn_mask=mask,
                                               scale=hd ** -0.5).squeeze(0)
            a = a.transpose(0, 1).reshape(T, c.q_heads * hd)
            h = h + F.linear(a, self.w(p + "self_attn.o_proj.weight"))
            x = rms_norm(h, self.w(p + "post_attention_layernorm.weight"), c.eps)
            gate = F.silu(F.linear(x, self.w(p + "mlp.gate_proj.weight")))
            x = F.linear(gate * F.linear(x, self.w(p + "mlp.up_proj.weight")), self.w(p + "mlp.down_proj.weight"))
            h = h + x
            if l == 0 and T > 1:
                self.trace["dec_layer0"] = h
        return rms_norm(h, self.w("model.language_model.norm.weight"), c.eps)

    def lm_head(self, h: torch.Tensor) -> torch.Tensor:
        name = "model.language_model.embed_tokens.weight" if self.c.tie or "lm_head.weight" not in self.sd else "lm_head.weight"
        return F.linear(h, self.w(name))

    def prefill(self, input_ids: torch.Tensor, pixel_values: torch.Tensor, grids):
        """Splice + full-prompt forward (HF modeling_qwen2_vl.py:1185-1253).  Returns logits [T, V], cache, delta."""
        c = self.c
        emb = F.embedding(input_ids, self.w("model.language_model.embed_tokens.weight"))
        img = self.vision(pixel_values, grids)
        mask = input_ids == c.image_token_id
        assert int(mask.sum()) == img.shape[0], "image features and image tokens do not match"
        emb = emb.clone()
        emb[mask] = img.to(emb.dtype)
        self.trace["inputs_embeds"] = emb
        pos3, delta = rope_index(input_ids, c.image_token_id, grids, c.merge)
        self.trace["position_ids"] = pos3
        cache = [None] * c.layers
        hn = self.decoder(emb, pos3, cache)
        return self.lm_head(hn), cache, delta

    def step(self, token: int, cache: list, delta: int) -> torch.Tensor:
        """One decode iteration (HF generation/utils.py:2876-2941 body; positions HF modeling_qwen2_vl.py:1167-1177)."""
        past = cache[0][0].shape[1]
        emb = F.embedding(torch.tensor([token]), self.w("model.language_model.embed_tokens.weight"))
        pos3 = torch.full((3, 1), past + delta, dtype=torch.long)
        return self.lm_head(self.decoder(emb, pos3, cache))[0]

    def generate(self, input_ids, pixel_values, grids, max_new: int, min_new: int = 0, forced: list | None = None,
                 repetition_penalty: float = 1.0):
        """Greedy loop (HF generation/utils.py:2783-2973 with do_sample=False): argmax of the fp32 copy of the last
        logits, EOS suppressed below min_new, stop at EOS / max_new.  `forced` teacher-forces the fed tokens while
        still recording every step's logits.  `repetition_penalty`: HF generation/logits_process.py
        RepetitionPenaltyLogitsProcessor over prompt + fed tokens.  Returns (tokens, per-step logits [n, V])."""
        c = self.c
        logits, cache, delta = self.prefill(input_ids, pixel_values, grids)
        last = logits[-1]
        toks, steps = [], []
        seen = set(int(t) for t in input_ids.tolist())
        self.processed_scores = []  # fp32 scores after the logits processors: what HF returns as `scores`
        for n in range(max_new):
            lf = last.float().clone()
            steps.append(last)
            if repetition_penalty != 1.0:
                idx = torch.tensor(sorted(seen))
                sc = lf[idx]
                lf[idx] = torch.where(sc < 0, sc * repetition_penalty, sc / repetition_penalty)
            if n < min_new:
                lf[list(c.eos_ids)] = -float("inf")
            self.processed_scores.append(lf)
            t = int(torch.argmax(lf))
            toks.append(t)
            fed = forced[n] if forced is not None else t
            seen.add(int(fed))
            if forced is None and t in c.eos_ids:
                break
            if n + 1 < max_new:
                last = self.step(fed, cache, delta)
        return toks, torch.stack(steps)
