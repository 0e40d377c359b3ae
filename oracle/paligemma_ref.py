"""ORACLE — test infrastructure, not product code.

CPU restatement (plain torch ops, batch 1, no transformers import) of what the reference's ``run_ocr`` executes
(``model.generate``, ocr_agent/tools.py:764-765) when OCR_MODEL is a PaliGemma checkpoint (BASELINE config 4): SigLIP
vision tower + linear projector + Gemma decoder with a bidirectional image+prompt prefix.  The arithmetic lives in the
third-party library transformers (reference pins 5.1.0, poetry.lock:5252-5253; validated here against the installed
5.15.0); every function cites the HF file:line it follows (HF = site-packages/transformers/models).

Pinned by tests/golden/paligemma_tiny_*.safetensors: outputs of the real HF classes on a seeded random-init model
(vision head_dim 72 and decoder head_dim 256 like the 3B checkpoint), written by tools/make_goldens.py; the product path
(ReadEngine, family "paligemma") is checked against them in tests/test_model_paligemma_gpu.py.

Only tests/ may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class PaliRefConfig:
    # SigLIP tower (HF siglip/configuration_siglip.py; PaliGemma-3B: 27 layers, 1152 wide, 16 heads, mlp 4304)
    v_layers: int = 27
    v_hidden: int = 1152
    v_heads: int = 16
    v_inter: int = 4304
    patch_size: int = 14
    image_size: int = 896
    v_eps: float = 1e-6
    # Gemma decoder (HF gemma/configuration_gemma.py; 2B: 18 layers, 2048 wide, 8 q / 1 kv heads of 256, mlp 16384)
    hidden: int = 2048
    layers: int = 18
    q_heads: int = 8
    kv_heads: int = 1
    head_dim: int = 256
    inter: int = 16384
    vocab: int = 257152
    rope_theta: float = 10000.0
    eps: float = 1e-6
    image_token_id: int = 257152
    eos_ids: tuple = (1,)
    pad_id: int = 0


def gemma_rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """HF gemma/modeling_gemma.py:64-79: everything in fp32, (1 + w), ONE cast at the end."""
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (xf * (1.0 + w.float())).type_as(x)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class PaliGemmaRef:
    """Functional restatement over a state dict with HF 5.x parameter names (model.vision_tower.*, model.language_model.*)."""

    def __init__(self, cfg: PaliRefConfig, sd: dict):
        self.c = cfg
        self.sd = sd
        self.dtype = sd["model.language_model.embed_tokens.weight"].dtype
        self.trace: dict = {}

    def w(self, name: str) -> torch.Tensor:
        return self.sd[name]

    # ------------------------------------------------------------------------------------------ vision tower
    def vision(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values fp32 [3, S, S] -> projected image features [P, hidden].  HF siglip/modeling_siglip.py:172-183
        (Conv2d patch embedding with bias + learned positions), :340-356 (pre-LN encoder layer), :268-300 (attention:
        separate q/k/v/out projections with bias, scale head_dim^-0.5, no mask), :310-322 (fc1 -> gelu_pytorch_tanh ->
        fc2), post_layernorm, then paligemma/modeling_paligemma.py:90-98 (linear projector)."""
        c = self.c
        v = "model.vision_tower."
        g = c.image_size // c.patch_size
        x = F.conv2d(pixel_values.to(self.dtype).unsqueeze(0), self.w(v + "embeddings.patch_embedding.weight"),
                     self.w(v + "embeddings.patch_embedding.bias"), stride=c.patch_size)
        x = x.flatten(2).transpose(1, 2)[0]                       # [P, d], raster order
        x = x + self.w(v + "embeddings.position_embedding.weight")
        self.trace["patch_embed"] = x
        P, hd = g * g, c.v_hidden // c.v_heads
        for l in range(c.v_layers):
            b = f"{v}encoder.layers.{l}."
            y = F.layer_norm(x, (c.v_hidden,), self.w(b + "layer_norm1.weight"), self.w(b + "layer_norm1.bias"), c.v_eps)
            q, k, vv = (F.linear(y, self.w(b + f"self_attn.{n}_proj.weight"), self.w(b + f"self_attn.{n}_proj.bias"))
                        .view(P, c.v_heads, hd).transpose(0, 1).unsqueeze(0) for n in ("q", "k", "v"))
            a = F.scaled_dot_product_attention(q, k, vv, scale=hd ** -0.5)[0].transpose(0, 1).reshape(P, c.v_hidden)
            x = x + F.linear(a, self.w(b + "self_attn.out_proj.weight"), self.w(b + "self_attn.out_proj.bias"))
            y = F.layer_norm(x, (c.v_hidden,), self.w(b + "layer_norm2.weight"), self.w(b + "layer_norm2.bias"), c.v_eps)
            y = F.gelu(F.linear(y, self.w(b + "mlp.fc1.weight"), self.w(b + "mlp.fc1.bias")), approximate="tanh")
            x = x + F.linear(y, self.w(b + "mlp.fc2.weight"), self.w(b + "mlp.fc2.bias"))
            if l == 0:
                self.trace["vit_block0"] = x
        x = F.layer_norm(x, (c.v_hidden,), self.w(v + "post_layernorm.weight"), self.w(v + "post_layernorm.bias"), c.v_eps)
        self.trace["vit_last"] = x
        y = F.linear(x, self.w("model.multi_modal_projector.linear.weight"), self.w("model.multi_modal_projector.linear.bias"))
        self.trace["projector"] = y
        return y

    # ------------------------------------------------------------------------------------------ decoder
    def _cos_sin(self, pos: torch.Tensor):
        """HF gemma/modeling_gemma.py:136-155: fp32 angles from fp32 inv_freq, cos/sin cast to the model dtype."""
        hd = self.c.head_dim
        inv = 1.0 / (self.c.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        fr = pos.float().unsqueeze(-1) * inv
        emb = torch.cat((fr, fr), dim=-1)
        return emb.cos().to(self.dtype), emb.sin().to(self.dtype)

    def decoder(self, h: torch.Tensor, pos: torch.Tensor, cache: list, bidirectional: bool):
        """Gemma layers + final norm (HF gemma/modeling_gemma.py:305-349, :240-302); h [T, hidden], pos [T] (1-indexed,
        paligemma/modeling_paligemma.py:237).  `bidirectional`: the T new rows form the image+prompt prefix and see each
        other (:256-262 with token_type_ids == 0); otherwise they are generated tokens and see the whole past."""
        c = self.c
        T, hd = h.shape[0], c.head_dim
        cos, sin = self._cos_sin(pos)
        cos, sin = cos.unsqueeze(0), sin.unsqueeze(0)
        for l in range(c.layers):
            p = f"model.language_model.layers.{l}."
            x = gemma_rms_norm(h, self.w(p + "input_layernorm.weight"), c.eps)
            q = F.linear(x, self.w(p + "self_attn.q_proj.weight")).view(T, c.q_heads, hd).transpose(0, 1)
            k = F.linear(x, self.w(p + "self_attn.k_proj.weight")).view(T, c.kv_heads, hd).transpose(0, 1)
            v = F.linear(x, self.w(p + "self_attn.v_proj.weight")).view(T, c.kv_heads, hd).transpose(0, 1)
            q = (q * cos) + (rotate_half(q) * sin)
            k = (k * cos) + (rotate_half(k) * sin)
            past = 0
            if cache[l] is not None:
                past = cache[l][0].shape[1]
                k = torch.cat([cache[l][0], k], dim=1)
                v = torch.cat([cache[l][1], v], dim=1)
            cache[l] = (k, v)
            g = c.q_heads // c.kv_heads
            kk, vv = k.repeat_interleave(g, dim=0), v.repeat_interleave(g, dim=0)
            mask = None
            if T > 1 and not bidirectional:
                mask = torch.ones(T, past + T, dtype=torch.bool).tril(past)
            a = F.scaled_dot_product_attention(q.unsqueeze(0), kk.unsqueeze(0), vv.unsqueeze(0), attn_mask=mask,
                                               scale=hd ** -0.5).squeeze(0)
            h = h + F.linear(a.transpose(0, 1).reshape(T, c.q_heads * hd), self.w(p + "self_attn.o_proj.weight"))
            x = gemma_rms_norm(h, self.w(p + "post_attention_layernorm.weight"), c.eps)
            gate = F.gelu(F.linear(x, self.w(p + "mlp.gate_proj.weight")), approximate="tanh")
            h = h + F.linear(gate * F.linear(x, self.w(p + "mlp.up_proj.weight")), self.w(p + "mlp.down_proj.weight"))
            if l == 0 and T > 1:
                self.trace["dec_layer0"] = h
        return gemma_rms_norm(h, self.w("model.language_model.norm.weight"), c.eps)

    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        """HF gemma/modeling_gemma.py:50-61: embedding rows times sqrt(hidden) rounded to the model dtype first."""
        scale = torch.tensor(self.c.hidden ** 0.5).to(self.dtype)
        return F.embedding(ids, self.w("model.language_model.embed_tokens.weight")) * scale

    def lm_head(self, h: torch.Tensor) -> torch.Tensor:
        return F.linear(h, self.w("model.language_model.embed_tokens.weight"))  # tied (paligemma :291)

    def prefill(self, input_ids: torch.Tensor, pixel_values: torch.Tensor):
        """Splice + prefix forward (HF paligemma/modeling_paligemma.py:222-275).  Returns logits [T, V] and the cache."""
        c = self.c
        ids = input_ids.clone()
        mask = ids == c.image_token_id
        if c.image_token_id >= c.vocab:
            ids[mask] = 0
        emb = self.embed(ids)
        img = self.vision(pixel_values)
        assert int(mask.sum()) == img.shape[0], "image features and image tokens do not match"
        emb[mask] = img.to(emb.dtype)
        self.trace["inputs_embeds"] = emb
        cache = [None] * c.layers
        pos = torch.arange(len(ids)) + 1
        return self.lm_head(self.decoder(emb, pos, cache, bidirectional=True)), cache

    def step(self, token: int, cache: list) -> torch.Tensor:
        past = cache[0][0].shape[1]
        h = self.decoder(self.embed(torch.tensor([token])), torch.tensor([past + 1]), cache, bidirectional=False)
        return self.lm_head(h)[0]

    def generate(self, input_ids, pixel_values, max_new: int, min_new: int = 0, forced: list | None = None):
        """Greedy loop as oracle/qwen2vl_ref.py (HF generation/utils.py:2783-2973 with do_sample=False)."""
        logits, cache = self.prefill(input_ids, pixel_values)
        last = logits[-1]
        toks, steps = [], []
        for n in range(max_new):
            lf = last.float().clone()
            steps.append(last)
            if n < min_new:
                lf[list(self.c.eos_ids)] = -float("inf")
            t = int(torch.argmax(lf))
            toks.append(t)
            fed = forced[n] if forced is not None else t
            if forced is None and t in self.c.eos_ids:
                break
            if n + 1 < max_new:
                last = self.step(fed, cache)
        return toks, torch.stack(steps)
