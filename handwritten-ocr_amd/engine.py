"""Read engine: the MI355X-native replacement for `model.generate` on the reference's OCR path
(ocr_agent/tools.py:764-765) for the Qwen2-VL family and the Qwen2.5-VL family (olmOCR-2, the reference's default
OCR_MODEL, ocr_agent/config.py:15).

Python here is plumbing only — shapes, index tables, buffer ownership (torch-ROCm tensors) and three calls into
libhwocr_hip.so per batch of reads:

    hwocr_vit_forward   page pixels -> image embeddings           (vision tower, batched over pages)
    hwocr_prefill       prompt with spliced image embeddings -> KV cache + first token
    hwocr_decode_graph  one captured HIP graph replayed once per generated token, all reads in lockstep

Weights stay resident in HBM for the life of the engine (the reference reloads the checkpoint after every
unload_ocr_model(), nodes.py:127,265); a "read" is one (page, preprocessing strategy) pair and the batch dimension
of every decode step is the number of reads in flight, so the decoder weights stream from HBM once per step for
all of them.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib, imageproc



@dataclass
class ModelConfig:
    name: str = "qwen2-vl-2b"
    # vision tower
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: float = 4.0
    patch_size: int = 14
    merge: int = 2
    tps: int = 2
    # Qwen2.5-VL tower only (family "qwen2_5_vl"): gated MLP width, attention window (pixels), full-attention layers
    family: str = "qwen2_vl"
    vit_inter: int = 0
    window_size: int = 112
    fullatt: tuple = (7, 15, 23, 31)
    # PaliGemma only (family "paligemma"): SigLIP sees a fixed square image; bos token id of the prompt
    image_size: int = 0
    bos_id: int = 2
    # decoder
    head_dim: int = 128   # 128 (Qwen families) or 256 (Gemma)
    hidden: int = 1536
    layers: int = 28
    q_heads: int = 12
    kv_heads: int = 2
    inter: int = 8960
    vocab: int = 151936
    rope_theta: float = 1_000_000.0
    mrope_section: tuple = (16, 24, 24)
    eps: float = 1e-6
    tie: bool = True
    # special tokens
    image_token_id: int = 151655
    vision_start_id: int = 151652
    vision_end_id: int = 151653
    im_start_id: int = 151644
    im_end_id: int = 151645
    eos_ids: tuple = (151645, 151643)
    pad_id: int = 151643
    # generation defaults of the checkpoint (generation_config.json): only the deterministic part is implemented — the
    # repetition penalty; temperatures of the Qwen-VL checkpoints (1e-6 .. 0.01 with top_k 1) make sampling the argmax
    repetition_penalty: float = 1.0
    sampling_note: str = ""  # what generation_config.json said about sampling (see _apply_generation_config)
    do_sample: bool = False  # generate(do_sample=True): hwocr_sample_advance ("hwocr sampling v1": DESIGN.md §2, oracle/sampling.py)
    temperature: float = 1.0
    top_k: int = 0           # 0 = off
    top_p: float = 1.0       # 1 = off
    # processor bounds (ocr_agent/config.py:17-18)
    min_pixels: int = 256 * 256
    max_pixels: int = 1024 * 1024

    @property
    def vit_hd(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def vit_hd_pad(self) -> int:
        """Width of a head inside the attention kernels: SigLIP's 72 is zero-padded to 80 (scores / outputs unchanged)."""
        return 80 if self.vit_hd == 72 else self.vit_hd

    @property
    def mlp_dim(self) -> int:
        """Width of the tower's MLP activation buffer; the Qwen2.5-VL intermediate size (3420) is zero-padded to x64."""
        if self.family in ("qwen2_5_vl", "paligemma"):
            return (self.vit_inter + 63) // 64 * 64
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def patch_k(self) -> int:
        return 3 * self.tps * self.patch_size ** 2

    @property
    def kpad(self) -> int:
        return (self.patch_k + 63) // 64 * 64

    def validate(self) -> None:
        pg = self.family == "paligemma"
        if self.head_dim not in (128, 256) or (not pg and self.hidden // self.q_heads != self.head_dim):
            raise ValueError("decoder head_dim must be 128 (or 256 for Gemma)")
        if self.vit_hd_pad not in (32, 64, 80, 128):
            raise ValueError("vision head_dim must be one of 32/64/72/80/128")
        if self.embed_dim % 64 or self.hidden % 64 or self.inter % 64 or self.vocab % 32 or self.mlp_dim % 64:
            raise ValueError("widths must be multiples of 64 (vocab: 32)")
        if not pg and sum(self.mrope_section) * 2 != self.head_dim:
            raise ValueError("mrope_section must sum to head_dim/2")
        if self.family not in ("qwen2_vl", "qwen2_5_vl", "paligemma"):
            raise ValueError(f"unknown model family {self.family!r}")
        if pg and (self.merge != 1 or self.tps != 1 or self.image_size % self.patch_size or self.vit_inter <= 0
                   or self.q_heads // self.kv_heads > 8):
            raise ValueError("paligemma needs merge == tps == 1, image_size a multiple of the patch, vit_inter, <= 8 q heads per kv head")
        if self.family == "qwen2_5_vl" and (self.vit_inter <= 0 or self.window_size < self.merge * self.patch_size):
            raise ValueError("qwen2_5_vl needs vit_inter and a window of at least one merged token")
        if self.hidden > 4096 or self.embed_dim > 4096:
            raise ValueError("norm kernels are built for rows of at most 4096 elements")


def preset(name: str) -> ModelConfig:
    """Shapes of BASELINE.json's configs (public model-card values)."""
    if name in ("qwen2-vl-2b", "Qwen/Qwen2-VL-2B-Instruct"):
        return ModelConfig()
    if name in ("qwen2.5-vl-7b", "olmocr-2-7b", "allenai/olmOCR-2-7B-1025", "Qwen/Qwen2.5-VL-7B-Instruct"):
        return ModelConfig(name="qwen2.5-vl-7b", family="qwen2_5_vl", vit_inter=3420, hidden=3584, layers=28, q_heads=28,
                           kv_heads=4, inter=18944, vocab=152064, tie=False)
    if name in ("qwen2.5-vl-3b", "Qwen/Qwen2.5-VL-3B-Instruct"):
        return ModelConfig(name="qwen2.5-vl-3b", family="qwen2_5_vl", vit_inter=3420, hidden=2048, layers=36, q_heads=16,
                           kv_heads=2, inter=11008, vocab=151936, tie=True)
    if name in ("paligemma-3b", "google/paligemma-3b-mix-896"):  # BASELINE config 4 (bf16; ReadEngine(fp8=True): E4M3 wide GEMMs)
        return ModelConfig(name="paligemma-3b", family="paligemma", depth=27, embed_dim=1152, num_heads=16, vit_inter=4304,
                           merge=1, tps=1, image_size=896, head_dim=256, hidden=2048, layers=18, q_heads=8, kv_heads=1,
                           inter=16384, vocab=257216, rope_theta=10000.0, tie=True, image_token_id=257152, eos_ids=(1,),
                           pad_id=0, bos_id=2)
    if name == "tinypg":  # tests/golden/paligemma_tiny.json
        return ModelConfig(name="tinypg", family="paligemma", depth=2, embed_dim=576, num_heads=8, vit_inter=600, merge=1,
                           tps=1, image_size=56, head_dim=256, hidden=256, layers=2, q_heads=2, kv_heads=1, inter=512,
                           vocab=512, rope_theta=10000.0, tie=True, image_token_id=500, eos_ids=(1,), pad_id=0, bos_id=2)
    if name == "tiny25":  # tests/golden/qwen25vl_tiny.json
        return ModelConfig(name="tiny25", family="qwen2_5_vl", depth=3, embed_dim=64, num_heads=2, vit_inter=88, window_size=56,
                           fullatt=(1,), hidden=256, layers=2, q_heads=2, kv_heads=1, inter=256, vocab=512, image_token_id=500,
                           vision_start_id=502, vision_end_id=503, im_start_id=504, im_end_id=505, eos_ids=(510,), pad_id=511,
                           min_pixels=28 * 28)
    if name == "tiny":  # tests/golden/qwen2vl_tiny.json
        return ModelConfig(name="tiny", depth=2, embed_dim=64, num_heads=2, mlp_ratio=2, hidden=256, layers=2, q_heads=2,
                           kv_heads=1, inter=256, vocab=512, image_token_id=500, vision_start_id=502, vision_end_id=503,
                           im_start_id=504, im_end_id=505, eos_ids=(510,), pad_id=511, min_pixels=28 * 28)
    if name == "small":  # BASELINE config 1: ViT-S-like + ~125M decoder (head_dim fixed to 128 by the kernels)
        return ModelConfig(name="small", depth=12, embed_dim=384, num_heads=6, mlp_ratio=4, hidden=768, layers=12,
                           q_heads=6, kv_heads=2, inter=3072, vocab=32768, image_token_id=32000, vision_start_id=32001,
                           vision_end_id=32002, im_start_id=32003, im_end_id=32004, eos_ids=(32004,), pad_id=32005)
    raise ValueError(f"unknown model preset {name!r}")


def random_state_dict(cfg: ModelConfig, seed: int = 0, device="cuda", std: float = 0.02) -> dict:
    """Random-init weights with HF parameter names (no checkpoint is reachable offline)."""
    g = torch.Generator(device=device).manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(torch.bfloat16)

    def ones_ish(n):
        return (1.0 + 0.02 * torch.randn(n, generator=g, device=device)).to(torch.bfloat16)

    sd = {}
    D, H = cfg.embed_dim, cfg.hidden
    HD = cfg.head_dim
    if cfg.family == "paligemma":
        v = "model.vision_tower."
        side = cfg.image_size // cfg.patch_size
        sd[v + "embeddings.patch_embedding.weight"], sd[v + "embeddings.patch_embedding.bias"] = rn(D, 3, cfg.patch_size, cfg.patch_size), rn(D)
        sd[v + "embeddings.position_embedding.weight"] = rn(side * side, D)
        for l in range(cfg.depth):
            b = f"{v}encoder.layers.{l}."
            for n in ("layer_norm1", "layer_norm2"):
                sd[b + n + ".weight"], sd[b + n + ".bias"] = ones_ish(D), rn(D)
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[b + f"self_attn.{n}.weight"], sd[b + f"self_attn.{n}.bias"] = rn(D, D), rn(D)
            sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"] = rn(cfg.vit_inter, D), rn(cfg.vit_inter)
            sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"] = rn(D, cfg.vit_inter), rn(D)
        sd[v + "post_layernorm.weight"], sd[v + "post_layernorm.bias"] = ones_ish(D), rn(D)
        sd["model.multi_modal_projector.linear.weight"], sd["model.multi_modal_projector.linear.bias"] = rn(H, D), rn(H)
        t = "model.language_model."
        sd[t + "embed_tokens.weight"] = rn(cfg.vocab, H)
        for l in range(cfg.layers):
            p = f"{t}layers.{l}."
            sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = rn(H), rn(H)  # (1 + w) norms
            sd[p + "self_attn.q_proj.weight"] = rn(cfg.q_heads * HD, H)
            sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = rn(cfg.kv_heads * HD, H), rn(cfg.kv_heads * HD, H)
            sd[p + "self_attn.o_proj.weight"] = rn(H, cfg.q_heads * HD)
            sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = rn(cfg.inter, H), rn(cfg.inter, H)
            sd[p + "mlp.down_proj.weight"] = rn(H, cfg.inter)
        sd[t + "norm.weight"] = rn(H)
        return sd
    v = "model.visual."
    sd[v + "patch_embed.proj.weight"] = rn(D, 3, cfg.tps, cfg.patch_size, cfg.patch_size)
    v25 = cfg.family == "qwen2_5_vl"
    for l in range(cfg.depth):
        b = f"{v}blocks.{l}."
        sd[b + "norm1.weight"], sd[b + "norm2.weight"] = ones_ish(D), ones_ish(D)
        sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"] = rn(3 * D, D), rn(3 * D)
        sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"] = rn(D, D), rn(D)
        if v25:
            I = cfg.vit_inter
            sd[b + "mlp.gate_proj.weight"], sd[b + "mlp.gate_proj.bias"] = rn(I, D), rn(I)
            sd[b + "mlp.up_proj.weight"], sd[b + "mlp.up_proj.bias"] = rn(I, D), rn(I)
            sd[b + "mlp.down_proj.weight"], sd[b + "mlp.down_proj.bias"] = rn(D, I), rn(D)
        else:
            sd[b + "norm1.bias"], sd[b + "norm2.bias"] = rn(D), rn(D)
            sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"] = rn(cfg.mlp_dim, D), rn(cfg.mlp_dim)
            sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"] = rn(D, cfg.mlp_dim), rn(D)
    MD = D * cfg.merge ** 2
    sd[v + "merger.ln_q.weight"] = ones_ish(D)
    if not v25:
        sd[v + "merger.ln_q.bias"] = rn(D)
    sd[v + "merger.mlp.0.weight"], sd[v + "merger.mlp.0.bias"] = rn(MD, MD), rn(MD)
    sd[v + "merger.mlp.2.weight"], sd[v + "merger.mlp.2.bias"] = rn(H, MD), rn(H)
    t = "model.language_model."
    sd[t + "embed_tokens.weight"] = rn(cfg.vocab, H)
    for l in range(cfg.layers):
        p = f"{t}layers.{l}."
        sd[p + "input_layernorm.weight"] = ones_ish(H)
        sd[p + "post_attention_layernorm.weight"] = ones_ish(H)
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"] = rn(cfg.q_heads * HD, H), rn(cfg.q_heads * HD)
        sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"] = rn(cfg.kv_heads * HD, H), rn(cfg.kv_heads * HD)
        sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"] = rn(cfg.kv_heads * HD, H), rn(cfg.kv_heads * HD)
        sd[p + "self_attn.o_proj.weight"] = rn(H, cfg.q_heads * HD)
        sd[p + "mlp.gate_proj.weight"] = rn(cfg.inter, H)
        sd[p + "mlp.up_proj.weight"] = rn(cfg.inter, H)
        sd[p + "mlp.down_proj.weight"] = rn(H, cfg.inter)
    sd[t + "norm.weight"] = ones_ish(H)
    if not cfg.tie:
        sd["lm_head.weight"] = rn(cfg.vocab, H)
    return sd


def normalize_keys(sd: dict) -> dict:
    """Accept both checkpoint layouts: `visual.* / model.*` (original release) and `model.visual.* /
    model.language_model.*` (transformers >= 4.52)."""
    out = {}
    for k, v in sd.items():
        if k.startswith(("vision_tower.", "multi_modal_projector.", "language_model.model.")):  # PaliGemma, transformers < 4.52
            k = "model." + k.replace("vision_tower.vision_model.", "vision_tower.").replace("language_model.model.", "language_model.")
        elif k.startswith("model.vision_tower.vision_model."):
            k = k.replace("vision_tower.vision_model.", "vision_tower.")
        elif k.startswith("visual."):
            k = "model." + k
        elif k.startswith("model.") and not k.startswith(("model.visual.", "model.language_model.", "model.vision_tower.",
                                                          "model.multi_modal_projector.")):
            k = "model.language_model." + k[len("model."):]
        out[k] = v
    return out


def _apply_generation_config(cfg: ModelConfig, path: str) -> None:
    """generation_config.json of the checkpoint = the defaults `model.generate(**inputs, max_new_tokens=...)` runs with in the
    reference (tools.py:765 passes nothing else).  Honoured: eos_token_id (int or list), pad_token_id, repetition_penalty,
    do_sample with temperature / top_k / top_p (HF's defaults 1.0 / 50 / 1.0 where the file is silent).  HF draws with
    torch.multinomial from torch's RNG stream (generation/utils.py:2919-2925), which no other implementation reproduces token for
    token: the draw here keeps HF's distribution but comes from a counter-based RNG keyed by (seed, read, step)
    (hwocr_sample_advance; seed = ReadEngine.seed / HWOCR_SAMPLE_SEED).  top_k == 1 is the argmax whatever the RNG says and stays on the
    greedy path (what the Qwen2-VL checkpoints ship)."""
    gen_path = os.path.join(path, "generation_config.json")
    if not os.path.exists(gen_path):
        return
    with open(gen_path) as f:
        g = json.load(f)
    cfg.repetition_penalty = float(g.get("repetition_penalty") or 1.0)
    eos = g.get("eos_token_id")
    if eos is not None:
        eos = tuple(int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos]))
        if len(eos) > 4:
            raise ValueError("generation_config.json lists more than 4 eos_token_id values (hwocr_gen_state.eos holds 4)")
        cfg.eos_ids = eos
    if g.get("pad_token_id") is not None:
        cfg.pad_id = int(g["pad_token_id"])
    if g.get("do_sample"):
        t = g.get("temperature")
        k = g.get("top_k")
        p = g.get("top_p")
        t, k, p = (1.0 if t is None else float(t)), (50 if k is None else int(k)), (1.0 if p is None else float(p))
        if k == 1:
            cfg.sampling_note = f"do_sample with top_k 1 (temperature {t}, top_p {p}): the argmax - greedy path"
        else:
            if not t > 0 or not p > 0:
                raise ValueError(f"generation_config.json: temperature {t} / top_p {p} must be positive")
            cfg.do_sample, cfg.temperature, cfg.top_k, cfg.top_p = True, t, k, p
            cfg.sampling_note = (f"do_sample: temperature {t}, top_k {k}, top_p {p} - drawn by hwocr_sample_advance (HF's distribution, "
                                 "own counter-based RNG: not HF's token stream)")


def load_checkpoint_dir(path: str, device="cuda") -> tuple[ModelConfig, dict]:
    """Load config.json + *.safetensors of a Qwen2-VL / Qwen2.5-VL (olmOCR-2) checkpoint directory (safetensors only;
    nothing is unpickled)."""
    from safetensors.torch import load_file

    with open(os.path.join(path, "config.json")) as f:
        hf = json.load(f)
    vc = hf.get("vision_config", {})
    tc = hf.get("text_config", hf)
    rope = tc.get("rope_parameters") or tc.get("rope_scaling") or {}
    if hf.get("model_type") == "paligemma":
        rope = tc.get("rope_parameters") or {}
        cfg = ModelConfig(
            name=os.path.basename(path.rstrip("/")), family="paligemma", depth=vc.get("num_hidden_layers", 27),
            embed_dim=vc.get("hidden_size", 1152), num_heads=vc.get("num_attention_heads", 16),
            vit_inter=vc.get("intermediate_size", 4304), patch_size=vc.get("patch_size", 14), merge=1, tps=1,
            image_size=vc.get("image_size", 896), head_dim=tc.get("head_dim", 256), hidden=tc.get("hidden_size", 2048),
            layers=tc.get("num_hidden_layers", 18), q_heads=tc.get("num_attention_heads", 8),
            kv_heads=tc.get("num_key_value_heads", 1), inter=tc.get("intermediate_size", 16384),
            vocab=tc.get("vocab_size", 257216), rope_theta=rope.get("rope_theta", tc.get("rope_theta", 10000.0)),
            eps=tc.get("rms_norm_eps", 1e-6), tie=True,
            image_token_id=hf.get("image_token_index", hf.get("image_token_id", 257152)),  # PaliGemma serialises `_index`
            eos_ids=(hf.get("eos_token_id", 1),), pad_id=hf.get("pad_token_id", 0), bos_id=hf.get("bos_token_id", 2))
        _apply_generation_config(cfg, path)
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, fn)))
        return cfg, {k: v.to(device=device, dtype=torch.bfloat16) for k, v in normalize_keys(sd).items()}
    v25 = hf.get("model_type") == "qwen2_5_vl" or vc.get("model_type") == "qwen2_5_vl_vision" or "fullatt_block_indexes" in vc
    tower = dict(family="qwen2_5_vl", embed_dim=vc.get("hidden_size", 1280), vit_inter=vc.get("intermediate_size", 3420),
                 window_size=vc.get("window_size", 112), fullatt=tuple(vc.get("fullatt_block_indexes", (7, 15, 23, 31)))) \
        if v25 else dict(embed_dim=vc.get("embed_dim", 1280), mlp_ratio=vc.get("mlp_ratio", 4))
    cfg = ModelConfig(
        name=os.path.basename(path.rstrip("/")), depth=vc.get("depth", 32), **tower,
        num_heads=vc.get("num_heads", 16), patch_size=vc.get("patch_size", 14),
        merge=vc.get("spatial_merge_size", 2), tps=vc.get("temporal_patch_size", 2), hidden=tc["hidden_size"],
        layers=tc["num_hidden_layers"], q_heads=tc["num_attention_heads"], kv_heads=tc["num_key_value_heads"],
        inter=tc["intermediate_size"], vocab=tc["vocab_size"], rope_theta=rope.get("rope_theta", tc.get("rope_theta", 1e6)),
        mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))), eps=tc.get("rms_norm_eps", 1e-6),
        tie=bool(hf.get("tie_word_embeddings", tc.get("tie_word_embeddings", False))),
        image_token_id=hf.get("image_token_id", 151655), vision_start_id=hf.get("vision_start_token_id", 151652),
        vision_end_id=hf.get("vision_end_token_id", 151653))
    _apply_generation_config(cfg, path)
    sd = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            sd.update(load_file(os.path.join(path, fn)))
    sd = {k: v.to(device=device, dtype=torch.bfloat16) for k, v in normalize_keys(sd).items()}
    return cfg, sd


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def interleave_rotary_pairs(t: torch.Tensor, heads: int, hd: int) -> torch.Tensor:
    """Rows of a q or k projection ([heads * hd, ...], or its bias [heads * hd]) reordered inside every head so that the two
    features a rotary pair is made of (d, d + hd/2) sit side by side: [d0, d0 + hd/2, d1, d1 + hd/2, ...].  The QKV GEMM of
    the vision tower rotates in its epilogue, where a lane owns 4 consecutive output features (hwocr_gemm_vit_qkv)."""
    tail = tuple(t.shape[1:])
    v = t.reshape((heads, 2, hd // 2) + tail)
    return v.transpose(1, 2).reshape((heads * hd,) + tail).contiguous()


def pick_attn_splits(reads: int, kv_heads: int) -> int:
    """Context splits of the decode attention: one 8-wave workgroup per (read, kv head) once those alone fill the chip,
    else 4-wave workgroups over 2..16 context splits + a merge launch."""
    return 1 if reads * kv_heads >= 160 else max(2, min(16, 768 // max(1, reads * kv_heads)))


DECODE_GEMMS = ("qkv", "o", "gate_up", "down", "lm_head")


def _placeholder_decoder(cfg: ModelConfig, fp8: bool = False):
    """A hwocr_decoder of `cfg`'s shape (one layer) whose pointers are placeholders: what the host-only entry points of the library
    (plan recording, hwocr_decode_slab_floats) need.  Returns (decoder, the placeholder pointer, objects to keep alive)."""
    layer = (_lib.DecLayer * 1)()
    one = C.c_void_p(64)  # non-NULL: the engine always binds one tiled copy of every decode weight
    for f in ("in_norm_w", "qkv_w", "o_w", "post_norm_w", "gate_up_w", "down_w"):
        setattr(layer[0], f, one)
    if cfg.family != "paligemma":
        layer[0].qkv_b = one
    if fp8:
        layer[0].qkv8t = layer[0].o8t = layer[0].gate_up8t = layer[0].down8t = one
        layer[0].qkv8 = layer[0].o8 = layer[0].gate_up8 = layer[0].down8 = _lib.W8(w=one, scale=one)
    else:
        layer[0].qkv_wt = layer[0].o_wt = layer[0].gate_up_wt = layer[0].down_wt = one
    dec = _lib.Decoder(layers=1, hidden=cfg.hidden, Hq=cfg.q_heads, Hkv=cfg.kv_heads, inter=cfg.inter, vocab=cfg.vocab, sec0=16, sec1=40,
                       head_dim=cfg.head_dim, gemma=1 if cfg.family == "paligemma" else 0, eps=cfg.eps, embed_scale=1.0, embed=one,
                       lm_head=one, final_norm_w=one, L=layer, rope_cos=one, rope_sin=one, max_pos=4096)
    if fp8:
        dec.lm_head8t = _lib.W8(w=one, scale=one)
    else:
        dec.lm_head_t = one
    return dec, one, (layer,)


def decode_slab_floats(cfg: ModelConfig, reads: int, fp8: bool = False) -> int:
    """fp32 elements a decode step at `reads` reads in flight writes into the workspace's split-K slab buffer
    (hwocr_decode_slab_floats: the size contract of hwocr_dec_ws.slabs).  Host-only."""
    dec, _, _keep = _placeholder_decoder(cfg, fp8)
    n = int(_lib.hip().hwocr_decode_slab_floats(C.byref(dec), reads))
    if n < 0:
        raise _lib.HwocrError(f"hwocr_decode_slab_floats refused reads={reads}")
    return n


def decode_plan(cfg: ModelConfig, reads: int, fp8: bool = False, attn_splits: int = 0) -> dict:
    """Kernel instances one decode step of `cfg` runs at `reads` reads in flight, as hwocr_decode_step itself lists them under plan
    recording (hwocr_plan_begin; one decoder layer + the LM head + the token selection; nothing is launched): {gemm name: (N, K, epi,
    splitk, variant)} for the five GEMMs, "attn": the attention instance, "launches": every line.  fp8: the engine was built with
    E4M3 decode weights (fp8 + fp8_decode).  Host-only."""
    import re

    lib = _lib.hip()
    dec, one, _keep = _placeholder_decoder(cfg, fp8)
    ws = _lib.DecWs(**{k: one for k in ("h", "hn", "qkv", "q", "attn", "act", "slabs", "part_o", "part_ml", "arrive", "select_ws", "logits")})
    kv = _lib.Kv(k=one, vt=one, nseq_max=max(reads, 1), ctx=2048, tiled=1 if cfg.head_dim == 128 else 0)
    if fp8 and cfg.head_dim == 256:   # the fp8 engine of a 256-wide-head model keeps an E4M3 KV cache (ReadEngine.fp8_kv)
        kv.k_scale, kv.v_scale, kv.fp8 = one, one, 1
    eos = (C.c_int * 4)(0, 0, 0, 0)
    gs = _lib.GenState(cur_ids=one, lens=one, n_gen=one, finished=one, out_tokens=one, rope_delta=one, max_new=8, min_new=0, n_eos=1,
                       pad_id=0, eos=eos, seen=None, seen_ld=0, rep_penalty=1.0, status=one, do_sample=0, temperature=1.0, top_k=0,
                       top_p=1.0, seed=0, read_ids=one)
    splits = attn_splits or pick_attn_splits(reads, cfg.kv_heads)
    _lib.check(lib.hwocr_plan_begin(), "hwocr_plan_begin")
    try:
        rc = lib.hwocr_decode_step(C.byref(dec), C.byref(ws), C.byref(kv), C.byref(gs), reads, splits, None)
    finally:
        need = C.c_int()
        lib.hwocr_plan_end(None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    _lib.check(rc, "hwocr_decode_step (plan)")
    lines = [l for l in buf.value.decode().split("\n") if l]
    gemms = [l for l in lines if l.startswith("gemm_")]
    if len(gemms) != len(DECODE_GEMMS):
        raise _lib.HwocrError(f"a decode step of one layer lists {len(gemms)} GEMM launches, expected {len(DECODE_GEMMS)}: {lines}")
    out = {"launches": lines}
    for name, line in zip(DECODE_GEMMS, gemms):
        variant, geom = line.split(" rows=", 1)
        f = dict(re.findall(r"(\w+)=(\d+)", "rows=" + geom))
        out[name] = (int(f["N"]), int(f["K"]), int(re.search(r"epi=(\d+)", variant).group(1)), int(f["splitk"]), variant)
    attn = [l for l in lines if l.startswith("attn_decode_kernel")]
    out["attn"] = attn[0].split(" ", 1)[0] if attn else ""
    return out


def wide_plan(cfg: ModelConfig, hw: tuple[int, int], pages: int, reads: int, prompt_len: int, fp8: bool = False) -> list[str]:
    """The launches of ONE hwocr_vit_forward over `pages` pages of hw = (H, W) pixels followed by ONE hwocr_prefill of `reads`
    prompts of `prompt_len` tokens, as the library itself lists them under plan recording (hwocr_plan_begin: every launcher checks
    its arguments, notes kernel instance + geometry and returns without touching the device) — so this is the launch sequence of
    csrc/runtime.hip, not a restatement of it.  Host-only: the structs carry placeholder pointers.  One line per launch."""
    lib = _lib.hip()
    one = C.c_void_p(64)
    w8 = _lib.W8(w=one, scale=one) if fp8 else _lib.W8()
    pg, v25 = cfg.family == "paligemma", cfg.family == "qwen2_5_vl"
    H, W = hw
    gh, gw = H // cfg.patch_size, W // cfg.patch_size
    P = gh * gw
    Pp = _ceil(P, 64)
    rows = pages * Pp
    hp = cfg.vit_hd_pad
    blocks = (_lib.VitBlock * cfg.depth)()
    for l in range(cfg.depth):
        for f in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"):
            setattr(blocks[l], f, one)
        blocks[l].windowed = 1 if (v25 and l not in cfg.fullatt) else 0
        D, DH = cfg.embed_dim, cfg.num_heads * hp
        # E4M3 copies exist where ReadEngine._w8 makes them: K a multiple of 128 and N of 8
        blocks[l].qkv8 = w8 if D % 128 == 0 else _lib.W8()
        blocks[l].proj8 = w8 if DH % 128 == 0 else _lib.W8()
        blocks[l].fc18 = w8 if D % 128 == 0 else _lib.W8()
        blocks[l].fc28 = w8 if cfg.mlp_dim % 128 == 0 else _lib.W8()
    vit = _lib.Vit(depth=cfg.depth, dim=cfg.embed_dim, heads=cfg.num_heads, mlp_dim=cfg.mlp_dim, patch=cfg.patch_size,
                   merge=cfg.merge, tps=cfg.tps, kpad=cfg.kpad, out_dim=cfg.hidden, kind=2 if pg else 1 if v25 else 0,
                   head_pad=hp if hp != cfg.vit_hd else 0, eps=1e-6, patch_w=one, blocks=blocks, merger_ln_w=one, merger_ln_b=one,
                   merger_fc1_w=one, merger_fc1_b=one, merger_fc2_w=one, merger_fc2_b=one, rope_cos=one, rope_sin=one,
                   pixel_lut=one, patch_b=one, pos_embed=one, qk_interleaved=1)
    ws = _lib.VitWs(**{k: one for k in ("patches", "x", "xn", "qkv", "q", "k", "vt", "attn", "mlp", "merge_mid")})
    if fp8:
        ws.q8, ws.q8s = one, one
    lay = _lib.VitLayout(pos_h=one, pos_w=one, seg_lens=one)
    if v25:
        _, win_lens = imageproc.window_order(gh, gw, cfg.merge, cfg.window_size, cfg.patch_size)
        lay.row_src = lay.win_off = lay.win_lens = one
        lay.nwin, lay.max_win = pages * len(win_lens), int(win_lens.max())
    layers = (_lib.DecLayer * cfg.layers)()
    HD = cfg.head_dim
    for l in range(cfg.layers):
        for f in ("in_norm_w", "qkv_w", "qkv_b", "o_w", "post_norm_w", "gate_up_w", "down_w", "qkv_wt", "o_wt", "gate_up_wt", "down_wt"):
            setattr(layers[l], f, one)
        if pg:
            layers[l].qkv_b = None
        layers[l].qkv8 = layers[l].gate_up8 = w8 if cfg.hidden % 128 == 0 else _lib.W8()
        layers[l].o8 = w8 if (cfg.q_heads * HD) % 128 == 0 else _lib.W8()
        layers[l].down8 = w8 if cfg.inter % 128 == 0 else _lib.W8()
    dec = _lib.Decoder(layers=cfg.layers, hidden=cfg.hidden, Hq=cfg.q_heads, Hkv=cfg.kv_heads, inter=cfg.inter, vocab=cfg.vocab,
                       sec0=16, sec1=40, head_dim=HD, gemma=1 if pg else 0, eps=cfg.eps, embed_scale=1.0, embed=one, lm_head=one,
                       lm_head_t=one, final_norm_w=one, L=layers, rope_cos=one, rope_sin=one, max_pos=4096)
    dws = _lib.DecWs(**{k: one for k in ("h", "hn", "qkv", "q", "attn", "act", "slabs", "part_o", "part_ml", "arrive", "select_ws", "logits")})
    if fp8:
        dws.q8, dws.q8s = one, one
    Tp = _ceil(prompt_len, 64)
    kv = _lib.Kv(k=one, vt=one, nseq_max=max(reads, 1), ctx=_ceil(Tp + 64, 64), tiled=1 if HD == 128 else 0)
    if fp8 and HD == 256:   # the E4M3 KV cache of the fp8 engine (ReadEngine.fp8_kv): the prefill fills it from a bf16 scratch
        kv.k_scale, kv.v_scale, kv.fp8 = one, one, 1
        dws.kt, dws.vtt = one, one
    eos = (C.c_int * 4)(0, 0, 0, 0)
    gs = _lib.GenState(cur_ids=one, lens=one, n_gen=one, finished=one, out_tokens=one, rope_delta=one, max_new=8, min_new=0,
                       n_eos=1, pad_id=0, eos=eos, seen=None, seen_ld=0, rep_penalty=1.0, status=one, do_sample=0,
                       temperature=1.0, top_k=0, top_p=1.0, seed=0, read_ids=one)
    _lib.check(lib.hwocr_plan_begin(), "hwocr_plan_begin")
    try:
        rc1 = lib.hwocr_vit_forward(C.byref(vit), C.byref(ws), one, pages, H, W, Pp, C.byref(lay), one, None) if pages > 0 else 0
        rc2 = lib.hwocr_prefill(C.byref(dec), C.byref(dws), C.byref(kv), C.byref(gs), one, one, one, one, one, one, reads, Tp, 0,
                                prompt_len, None) if reads > 0 else 0
    finally:
        need = C.c_int()
        lib.hwocr_plan_end(None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        lib.hwocr_plan_end(buf, len(buf), C.byref(need))
    _lib.check(rc1, "hwocr_vit_forward (plan)")
    _lib.check(rc2, "hwocr_prefill (plan)")
    return [l for l in buf.value.decode().split("\n") if l]


class _IndexView:
    """seq[idx[0]], seq[idx[1]], ... without touching any item before it is asked for."""

    def __init__(self, seq, idx: list[int]):
        self.seq, self.idx = seq, idx

    def __len__(self) -> int:
        return len(self.idx)

    def __getitem__(self, i: int):
        return self.seq[self.idx[i]]


class ReadSource:
    """The reads of one job that are still waiting for a decode slot, shared by the lanes that work on it: a lane whose slots free up
    TAKES the next reads (thread-safe), so the lanes stay balanced whatever the reads' lengths, and with reads of equal length the
    last, short round goes to one lane as one batch instead of leaving every lane half empty (a decode step costs almost the same
    at 132 rows as at 252: the weights stream either way)."""

    def __init__(self, n: int):
        import threading

        self.n = n
        self._next = 0
        self._lock = threading.Lock()

    def remaining(self) -> int:
        return self.n - self._next

    def take(self, k: int) -> list[int]:
        with self._lock:
            lo = self._next
            self._next = min(self.n, lo + max(0, k))
            return list(range(lo, self._next))


class ReadEngine:
    VIT_MAX_GRID = 2048  # rows of the vision rotary table: the longest page side in patches (smart_resize admits 200:1 strips:
    #                      sqrt(1024^2 * 200) / 14 = 1035 patches at the reference's max_pixels); encode_pages checks it
    MAX_GRAPHS = 8  # captured decode graphs kept (one per distinct reads-in-flight / generation setting), least recently used out

    def __init__(self, cfg: ModelConfig, state_dict: dict, max_reads: int = 96, ctx: int = 2048, device: str | None = None,
                 vit_batch: int = 8, prefill_batch: int = 16, attn_splits: int = 0, fp8: bool = False,
                 fp8_decode: bool | None = None, fp8_kv: bool | None = None):
        """device: None = the process's current device (one process per GPU: shard.init_from_env has already made
        LOCAL_RANK's device current).  fp8: run the wide GEMMs of the vision tower and of the decoder prefill (K a multiple
        of 128) on E4M3 copies of the weights with per-token activation scales (BASELINE config 4); norms and attention stay
        bf16.  fp8_decode (with fp8; default: HWOCR_FP8_DECODE, off): the decode GEMMs and the LM head also read E4M3 weight codes
        (hwocr_gemm_skinny_w8: half the weight bytes per step).  Exact, but measured SLOWER than the bf16 decode weights on this
        part (PaliGemma-3B, 252 reads: 6.08 vs 5.86 ms per token — the decode GEMMs are not weight-byte bound), hence opt-in."""
        cfg.validate()
        if not torch.cuda.is_available():
            raise _lib.HwocrError("ReadEngine needs an MI355X (ROCm) device: there is no CPU path")
        if max_reads < 1 or max_reads > 256:
            raise ValueError("max_reads must be in 1..256 (one decode batch)")
        if attn_splits < 0 or attn_splits > 16:
            raise ValueError("attn_splits must be 0 (automatic) or 1..16 (the partial buffers hold 16 splits)")
        self.cfg = cfg
        self.dev = torch.device(f"cuda:{torch.cuda.current_device()}" if device is None else device)
        torch.cuda.set_device(self.dev)
        self.lib = _lib.hip()
        self.max_reads = max_reads
        self.ctx = _ceil(ctx, 64)
        self.vit_batch = vit_batch
        self.prefill_batch = prefill_batch
        self.attn_splits = attn_splits or int(os.environ.get("HWOCR_ATTN_SPLITS", "0"))  # 0: pick_attn_splits
        self.fp8 = bool(fp8)
        # E4M3 KV cache (with fp8, 256-wide heads = PaliGemma / Gemma; default on, HWOCR_FP8_KV=0 or fp8_kv=False: bf16 cache): one byte
        # per cached element + one scale per token and kv head; the decode attention of config 4 streams half the bytes
        self.fp8_kv = self.fp8 and cfg.head_dim == 256 and (os.environ.get("HWOCR_FP8_KV", "1") not in ("", "0") if fp8_kv is None else bool(fp8_kv))
        self.fp8_decode = self.fp8 and (os.environ.get("HWOCR_FP8_DECODE", "0") not in ("", "0") if fp8_decode is None else bool(fp8_decode))
        self.collect_timings = False
        self.timings = {}
        self.seed = int(os.environ.get("HWOCR_SAMPLE_SEED", "0")) & (2 ** 64 - 1)  # key of the sampling RNG (cfg.do_sample reads only)
        self._keep = []  # everything the C structs point at
        self._tiled_keep = []
        self._graphs = {}
        self._bind_weights(normalize_keys(state_dict))
        self._alloc_state()

    # ------------------------------------------------------------------------------------------ weights
    def _t(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(device=self.dev, dtype=torch.bfloat16).contiguous()
        self._keep.append(x)
        return x

    def _tiled(self, w2d: torch.Tensor) -> torch.Tensor:
        n, k = w2d.shape
        out = torch.empty(n * k, dtype=torch.bfloat16, device=self.dev)
        _lib.check(self.lib.hwocr_tile_weights(_lib.ptr(w2d), _lib.ptr(out), n, k, k, _lib.stream_handle()),
                   "hwocr_tile_weights")
        self._tiled_keep.append(out)
        return out

    def _w8q(self, w2d: torch.Tensor):
        """(hwocr_w8 pack, E4M3 codes [N][K] or None) of a bound weight: hwocr_quant_rows_fp8 per output feature, or the empty pack.
        The codes tensor is returned beside the struct: a ctypes struct stored into a parent struct is copied field by field, so
        a Python attribute hung on it does not survive `L.qkv8 = pack; L.qkv8` (the decode re-tiling never saw it)."""
        n, k = w2d.shape
        if not self.fp8 or k % 128 or n % 8:
            return _lib.W8(), None
        q = torch.empty(n, k, dtype=torch.uint8, device=self.dev)
        s = torch.empty(n, dtype=torch.float32, device=self.dev)
        _lib.check(self.lib.hwocr_quant_rows_fp8(_lib.ptr(w2d), _lib.ptr(q), _lib.ptr(s), n, k, k, k, _lib.stream_handle()),
                   "hwocr_quant_rows_fp8")
        self._keep += [q, s]
        return _lib.W8(w=_lib.ptr(q), scale=_lib.ptr(s)), q

    def _w8(self, w2d: torch.Tensor) -> _lib.W8:
        return self._w8q(w2d)[0]

    def _w8_tiled(self, q: torch.Tensor | None):
        """Byte-tiled copy of an E4M3 weight's codes for the decode GEMMs (hwocr_tile_weights_fp8), or None."""
        if q is None or q.shape[0] % 16 or q.shape[1] % 64 or not self.fp8_decode:
            return None
        n, k = q.shape
        out = torch.empty(n * k, dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.hwocr_tile_weights_fp8(_lib.ptr(q), _lib.ptr(out), n, k, k, _lib.stream_handle()), "hwocr_tile_weights_fp8")
        self._tiled_keep.append(out)
        return _lib.ptr(out)

    def _bind_weights(self, sd: dict) -> None:
        if self.cfg.family == "paligemma":
            self._bind_vision_siglip(sd)
        else:
            self._bind_vision_qwen(sd)
        self._bind_decoder(sd)

    def _bind_vision_siglip(self, sd: dict) -> None:
        """SigLIP tower + projector (HF siglip/modeling_siglip.py:116-356, paligemma/modeling_paligemma.py:90-98) on the
        kernels of the Qwen towers: q/k/v fused into one GEMM whose head rows are zero-padded 72 -> 80 (out_proj gets
        zero columns), so the head-dim-80 attention kernel runs unchanged with scale 72^-1/2; no rotary (the rope kernel
        sees position 0 = identity); learned positions are the residual operand of the patch GEMM."""
        c, P, dev, bf = self.cfg, _lib.ptr, self.dev, torch.bfloat16
        v = "model.vision_tower."
        D, Hn, hd, hp = c.embed_dim, c.num_heads, c.vit_hd, c.vit_hd_pad
        pw = torch.zeros(D, c.kpad, dtype=bf, device=dev)
        pw[:, : c.patch_k] = sd[v + "embeddings.patch_embedding.weight"].reshape(D, -1).to(dev, bf)
        self._keep.append(pw)

        def pad_heads_rows(w):   # [Hn*hd, ...] -> [Hn*hp, ...]
            w = w.to(dev, bf)
            out = torch.zeros((Hn, hp) + tuple(w.shape[1:]), dtype=bf, device=dev)
            out[:, :hd] = w.reshape((Hn, hd) + tuple(w.shape[1:]))
            return out.reshape((Hn * hp,) + tuple(w.shape[1:]))

        def pad_rows(w, n):
            w = w.to(dev, bf)
            out = torch.zeros((n,) + tuple(w.shape[1:]), dtype=bf, device=dev)
            out[: w.shape[0]] = w
            return out

        blocks = (_lib.VitBlock * c.depth)()
        for l in range(c.depth):
            b = f"{v}encoder.layers.{l}."
            il = lambda t, n: interleave_rotary_pairs(t, Hn, hp) if n in "qk" else t  # noqa: E731  (pairs of the padded head)
            qkv_w = torch.cat([il(pad_heads_rows(sd[b + f"self_attn.{n}_proj.weight"]), n) for n in "qkv"], dim=0)
            qkv_b = torch.cat([il(pad_heads_rows(sd[b + f"self_attn.{n}_proj.bias"]), n) for n in "qkv"], dim=0)
            ow = sd[b + "self_attn.out_proj.weight"].to(dev, bf)                        # [D][Hn*hd]
            proj_w = torch.zeros(D, Hn, hp, dtype=bf, device=dev)
            proj_w[:, :, :hd] = ow.reshape(D, Hn, hd)
            fc2 = torch.zeros(D, c.mlp_dim, dtype=bf, device=dev)
            fc2[:, : c.vit_inter] = sd[b + "mlp.fc2.weight"].to(dev, bf)
            B = blocks[l]
            B.ln1_w, B.ln1_b = P(self._t(sd[b + "layer_norm1.weight"])), P(self._t(sd[b + "layer_norm1.bias"]))
            B.ln2_w, B.ln2_b = P(self._t(sd[b + "layer_norm2.weight"])), P(self._t(sd[b + "layer_norm2.bias"]))
            w_qkv, w_proj = self._t(qkv_w), self._t(proj_w.reshape(D, Hn * hp))
            w_fc1, w_fc2 = self._t(pad_rows(sd[b + "mlp.fc1.weight"], c.mlp_dim)), self._t(fc2)
            B.qkv_w, B.qkv_b = P(w_qkv), P(self._t(qkv_b))
            B.proj_w, B.proj_b = P(w_proj), P(self._t(sd[b + "self_attn.out_proj.bias"]))
            B.fc1_w, B.fc1_b = P(w_fc1), P(self._t(pad_rows(sd[b + "mlp.fc1.bias"], c.mlp_dim)))
            B.fc2_w, B.fc2_b = P(w_fc2), P(self._t(sd[b + "mlp.fc2.bias"]))
            B.qkv8, B.proj8, B.fc18, B.fc28 = self._w8(w_qkv), self._w8(w_proj), self._w8(w_fc1), self._w8(w_fc2)
        # identity rotation table for the (unused) rotary of the shared rope/split kernel: row 0 = (cos 1, sin 0)
        self.vit_cos = torch.ones(8, hp // 4, dtype=torch.float32, device=dev)
        self.vit_sin = torch.zeros(8, hp // 4, dtype=torch.float32, device=dev)
        self.lut = torch.from_numpy(imageproc.pixel_lut((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))).to(bf).to(dev)
        self.vit = _lib.Vit(depth=c.depth, dim=D, heads=Hn, mlp_dim=c.mlp_dim, patch=c.patch_size, merge=1, tps=1, kpad=c.kpad,
                            out_dim=c.hidden, kind=2, head_pad=hp if hp != hd else 0, eps=1e-6, patch_w=P(pw), blocks=blocks,
                            merger_ln_w=P(self._t(sd[v + "post_layernorm.weight"])), merger_ln_b=P(self._t(sd[v + "post_layernorm.bias"])),
                            merger_fc2_w=P(self._t(sd["model.multi_modal_projector.linear.weight"])),
                            merger_fc2_b=P(self._t(sd["model.multi_modal_projector.linear.bias"])),
                            rope_cos=P(self.vit_cos), rope_sin=P(self.vit_sin), pixel_lut=P(self.lut),
                            patch_b=P(self._t(sd[v + "embeddings.patch_embedding.bias"])),
                            pos_embed=P(self._t(sd[v + "embeddings.position_embedding.weight"])), qk_interleaved=1)
        self._keep.append(blocks)

    def _bind_vision_qwen(self, sd: dict) -> None:
        c = self.cfg
        P = _lib.ptr
        v = "model.visual."
        D = c.embed_dim
        pw = torch.zeros(D, c.kpad, dtype=torch.bfloat16, device=self.dev)
        pw[:, : c.patch_k] = sd[v + "patch_embed.proj.weight"].reshape(D, -1).to(self.dev, torch.bfloat16)
        self._keep.append(pw)
        blocks = (_lib.VitBlock * c.depth)()
        v25 = c.family == "qwen2_5_vl"
        for l in range(c.depth):
            b = f"{v}blocks.{l}."
            common = (("ln1_w", "norm1.weight"), ("qkv_w", "attn.qkv.weight"), ("qkv_b", "attn.qkv.bias"),
                      ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"), ("ln2_w", "norm2.weight"))
            v2 = (("ln1_b", "norm1.bias"), ("ln2_b", "norm2.bias"), ("fc1_w", "mlp.fc1.weight"), ("fc1_b", "mlp.fc1.bias"),
                  ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias"))
            bound = {}
            for fld, key in common + (() if v25 else v2):
                t = sd[b + key]
                if fld in ("qkv_w", "qkv_b"):  # q and k thirds: rotary pairs side by side (hwocr_vit.qk_interleaved)
                    Dq = t.shape[0] // 3
                    t = torch.cat([interleave_rotary_pairs(t[:Dq], c.num_heads, c.vit_hd),
                                   interleave_rotary_pairs(t[Dq: 2 * Dq], c.num_heads, c.vit_hd), t[2 * Dq:]], dim=0)
                bound[fld] = self._t(t)
                setattr(blocks[l], fld, P(bound[fld]))
            if v25:
                # gate/up (+ biases) zero-padded to mlp_dim rows and interleaved in 16-row tiles for the SwiGLU epilogue;
                # padded rows give silu(0) * 0 = 0 and meet zero columns of down_proj
                I, Ip = c.vit_inter, c.mlp_dim

                def padrows(t):
                    t = t.to(self.dev, torch.bfloat16)
                    out = torch.zeros((Ip,) + tuple(t.shape[1:]), dtype=torch.bfloat16, device=self.dev)
                    out[:I] = t
                    return out

                def interleave(g, u):
                    g, u = padrows(g), padrows(u)
                    tail = tuple(g.shape[1:])
                    return torch.stack([g.reshape((Ip // 16, 16) + tail), u.reshape((Ip // 16, 16) + tail)], dim=1) \
                        .reshape((2 * Ip,) + tail)

                down = torch.zeros(D, Ip, dtype=torch.bfloat16, device=self.dev)
                down[:, :I] = sd[b + "mlp.down_proj.weight"].to(self.dev, torch.bfloat16)
                bound["fc1_w"] = self._t(interleave(sd[b + "mlp.gate_proj.weight"], sd[b + "mlp.up_proj.weight"]))
                bound["fc2_w"] = self._t(down)
                blocks[l].fc1_w = P(bound["fc1_w"])
                blocks[l].fc1_b = P(self._t(interleave(sd[b + "mlp.gate_proj.bias"], sd[b + "mlp.up_proj.bias"])))
                blocks[l].fc2_w, blocks[l].fc2_b = P(bound["fc2_w"]), P(self._t(sd[b + "mlp.down_proj.bias"]))
                blocks[l].windowed = 0 if l in c.fullatt else 1
            blocks[l].qkv8, blocks[l].proj8 = self._w8(bound["qkv_w"]), self._w8(bound["proj_w"])
            blocks[l].fc18, blocks[l].fc28 = self._w8(bound["fc1_w"]), self._w8(bound["fc2_w"])
        # vision rotary table, fp32, exactly as the library builds it (positions * inv_freq, then cos/sin)
        hd = c.vit_hd
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
        ang = torch.arange(self.VIT_MAX_GRID, dtype=torch.float).unsqueeze(-1) * inv
        self.vit_cos = ang.cos().contiguous().to(self.dev)
        self.vit_sin = ang.sin().contiguous().to(self.dev)
        self.lut = torch.from_numpy(imageproc.pixel_lut()).to(torch.bfloat16).to(self.dev)
        self.vit = _lib.Vit(depth=c.depth, dim=D, heads=c.num_heads, mlp_dim=c.mlp_dim, patch=c.patch_size, merge=c.merge,
                            tps=c.tps, kpad=c.kpad, out_dim=c.hidden, kind=1 if v25 else 0, eps=1e-6, patch_w=P(pw),
                            blocks=blocks, merger_ln_w=P(self._t(sd[v + "merger.ln_q.weight"])),
                            merger_ln_b=None if v25 else P(self._t(sd[v + "merger.ln_q.bias"])),
                            merger_fc1_w=P(self._t(sd[v + "merger.mlp.0.weight"])), merger_fc1_b=P(self._t(sd[v + "merger.mlp.0.bias"])),
                            merger_fc2_w=P(self._t(sd[v + "merger.mlp.2.weight"])), merger_fc2_b=P(self._t(sd[v + "merger.mlp.2.bias"])),
                            rope_cos=P(self.vit_cos), rope_sin=P(self.vit_sin), pixel_lut=P(self.lut), qk_interleaved=1)
        self._keep.append(blocks)

    def _bind_decoder(self, sd: dict) -> None:
        c, P, HD = self.cfg, _lib.ptr, self.cfg.head_dim
        gemma = c.family == "paligemma"
        t = "model.language_model."
        layers = (_lib.DecLayer * c.layers)()
        for l in range(c.layers):
            p = f"{t}layers.{l}."
            qkv_w = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                               sd[p + "self_attn.v_proj.weight"]], dim=0)
            qkv_b = None if gemma else torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"],
                                                  sd[p + "self_attn.v_proj.bias"]], dim=0)
            # gate/up rows interleaved in 16-row tiles so one MFMA tile pair yields silu(gate)*up (csrc/gemm.hip)
            g = sd[p + "mlp.gate_proj.weight"].reshape(c.inter // 16, 16, c.hidden)
            u = sd[p + "mlp.up_proj.weight"].reshape(c.inter // 16, 16, c.hidden)
            gu = torch.stack([g, u], dim=1).reshape(2 * c.inter, c.hidden)
            L = layers[l]
            w_qkv, w_o = self._t(qkv_w), self._t(sd[p + "self_attn.o_proj.weight"])
            w_gu, w_down = self._t(gu), self._t(sd[p + "mlp.down_proj.weight"])
            L.in_norm_w = P(self._t(sd[p + "input_layernorm.weight"]))
            L.qkv_w, L.qkv_b = P(w_qkv), (None if qkv_b is None else P(self._t(qkv_b)))
            L.o_w = P(w_o)
            L.post_norm_w = P(self._t(sd[p + "post_attention_layernorm.weight"]))
            L.gate_up_w, L.down_w = P(w_gu), P(w_down)
            # decode copies: E4M3 codes byte-tiled for the streaming kernel where the layer has them (fp8_decode: half the weight
            # bytes per decode step), else bf16 in MFMA-fragment order (contiguous KiB per fragment load; 288 GB of HBM pays
            # for the copy)
            for name, w in (("qkv", w_qkv), ("o", w_o), ("gate_up", w_gu), ("down", w_down)):
                pack, codes = self._w8q(w)
                setattr(L, name + "8", pack)
                t8 = self._w8_tiled(codes)
                if t8 is not None:
                    setattr(L, name + "8t", t8)
                else:
                    setattr(L, name + "_wt", P(self._tiled(w)))
        embed = self._t(sd[t + "embed_tokens.weight"])
        head = embed if (c.tie or "lm_head.weight" not in sd) else self._t(sd["lm_head.weight"])
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, HD, 2, dtype=torch.float) / HD))
        ang = torch.arange(max(self.ctx + 64, 4096), dtype=torch.float).unsqueeze(-1) * inv
        self.dec_cos = ang.cos().to(torch.bfloat16).contiguous().to(self.dev)
        self.dec_sin = ang.sin().to(torch.bfloat16).contiguous().to(self.dev)
        self.max_pos = ang.shape[0]
        # Gemma: plain RoPE = every frequency on the first position axis (sec0 past the last frequency)
        sec0, sec1 = (HD, HD) if gemma else (c.mrope_section[0], c.mrope_section[0] + c.mrope_section[1])
        self.dec = _lib.Decoder(layers=c.layers, hidden=c.hidden, Hq=c.q_heads, Hkv=c.kv_heads, inter=c.inter, vocab=c.vocab,
                                sec0=sec0, sec1=sec1, head_dim=HD, gemma=1 if gemma else 0, eps=c.eps,
                                embed_scale=float(c.hidden) ** 0.5,
                                embed=P(embed), lm_head=P(head), final_norm_w=P(self._t(sd[t + "norm.weight"])), L=layers,
                                rope_cos=P(self.dec_cos), rope_sin=P(self.dec_sin), max_pos=int(ang.shape[0]))
        head8, head_codes = self._w8q(head) if self.fp8_decode else (_lib.W8(), None)
        head8t = self._w8_tiled(head_codes)
        if head8t is not None:
            self.dec.lm_head8t = _lib.W8(w=head8t, scale=head8.scale)
            self._keep.append(head8)
        else:
            self.dec.lm_head_t = P(self._tiled(head))
        self._keep.append(layers)
        self.embed_weight = embed

    # ------------------------------------------------------------------------------------------ buffers
    def _alloc_state(self) -> None:
        c, R, dev = self.cfg, self.max_reads, self.dev
        bf = torch.bfloat16
        HD = c.head_dim
        kv_elems = c.layers * R * c.kv_heads * self.ctx * HD
        if self.fp8_kv:   # E4M3 codes in operand order (csrc/common.h kv8_k / kv8_v) + scales [layer][read][kv head][ctx]
            self.k_cache = torch.zeros(kv_elems, dtype=torch.uint8, device=dev)
            self.vt_cache = torch.zeros(kv_elems, dtype=torch.uint8, device=dev)
            self.k_scale = torch.ones(kv_elems // HD, dtype=torch.float32, device=dev)
            self.v_scale = torch.ones(kv_elems // HD, dtype=torch.float32, device=dev)
            self.kv = _lib.Kv(k=_lib.ptr(self.k_cache), vt=_lib.ptr(self.vt_cache), nseq_max=R, ctx=self.ctx, tiled=0,
                              k_scale=_lib.ptr(self.k_scale), v_scale=_lib.ptr(self.v_scale), fp8=1)
        else:
            self.k_cache = torch.zeros(kv_elems, dtype=bf, device=dev)
            self.vt_cache = torch.zeros(kv_elems, dtype=bf, device=dev)
            # fragment-tiled cache for head_dim 128; Gemma's 256-wide heads use the row layout
            self.kv = _lib.Kv(k=_lib.ptr(self.k_cache), vt=_lib.ptr(self.vt_cache), nseq_max=R, ctx=self.ctx,
                              tiled=1 if HD == 128 else 0)
        i32 = dict(dtype=torch.int32, device=dev)
        self.cur_ids = torch.zeros(R, **i32)
        self.lens = torch.zeros(R, **i32)
        self.n_gen = torch.zeros(R, **i32)
        self.finished = torch.zeros(R, **i32)
        self.rope_delta = torch.zeros(R, **i32)
        self.status = torch.zeros(1, **i32)  # HWOCR_STATUS_* bits raised on the device (hwocr_gen_state.status)
        self.read_ids = torch.arange(R, **i32)  # the caller's number of the read in each slot (sampling RNG counter)
        self.out_tokens = None
        self._seen = None
        self._tok_bufs = {}
        self._ws_dec = None
        self._ws_rows = 0
        self._vit_rows = 0

    def _dec_ws(self, rows: int) -> _lib.DecWs:
        """Workspace for prefill (rows = reads x padded prompt length) and decode (rows = reads)."""
        c, dev, bf = self.cfg, self.dev, torch.bfloat16
        if self._ws_dec is None or rows > self._ws_rows:
            R = self.max_reads
            HD = c.head_dim
            QW = (c.q_heads + 2 * c.kv_heads) * HD
            slab_elems = 40 * R * max(QW, c.hidden)  # split-K never exceeds ceil(K/256) <= 35 slices here
            splits = 16
            self._bufs = dict(
                h=torch.empty(rows, c.hidden, dtype=bf, device=dev), hn=torch.empty(rows, c.hidden, dtype=bf, device=dev),
                qkv=torch.empty(rows, QW, dtype=bf, device=dev), q=torch.empty(rows, c.q_heads * HD, dtype=bf, device=dev),
                attn=torch.empty(rows, c.q_heads * HD, dtype=bf, device=dev), act=torch.empty(rows, c.inter, dtype=bf, device=dev),
                slabs=torch.empty(slab_elems, dtype=torch.float32, device=dev),
                part_o=torch.empty(R * c.q_heads * splits * HD, dtype=torch.float32, device=dev),
                part_ml=torch.empty(R * c.q_heads * splits * 2, dtype=torch.float32, device=dev),
                # arrival counters of the split decode attention / the split token selection: zero here, left zero by every launch
                arrive=torch.zeros(R * c.kv_heads, dtype=torch.int32, device=dev),
                select_ws=torch.zeros(R * _lib.SELECT_WS_INTS, dtype=torch.int32, device=dev),
                logits=torch.empty(R, c.vocab, dtype=bf, device=dev))
            if self.fp8:  # E4M3 staging of one prefill GEMM input + its row scales
                self._bufs["q8"] = torch.empty(rows * max(c.hidden, c.q_heads * HD, c.inter), dtype=torch.uint8, device=dev)
                self._bufs["q8s"] = torch.empty(rows, dtype=torch.float32, device=dev)
            if self.fp8_kv:  # one prefill call's K / V^T of one layer in bf16 (the prompt is attended over these; the cache gets codes)
                self._bufs["kt"] = torch.zeros(rows * c.kv_heads * HD + 64 * HD, dtype=bf, device=dev)
                self._bufs["vtt"] = torch.zeros(rows * c.kv_heads * HD + 64 * HD, dtype=bf, device=dev)
            for n in {1, min(16, R), min(17, R), R}:   # the library's own statement of what a step writes there (hwocr.h)
                need = int(self.lib.hwocr_decode_slab_floats(C.byref(self.dec), n))
                if need < 0 or need > slab_elems:
                    raise _lib.HwocrError(f"decode slab buffer: {slab_elems} fp32 allocated, a step at {n} reads writes {need}")
            self._ws_rows = rows
            self._ws_dec = _lib.DecWs(**{k: _lib.ptr(v) for k, v in self._bufs.items()})
            self._drop_graphs()  # graphs bake workspace pointers
        return self._ws_dec

    def _vit_ws(self, rows: int) -> _lib.VitWs:
        c, dev, bf = self.cfg, self.dev, torch.bfloat16
        if rows > self._vit_rows:
            D, DH = c.embed_dim, c.num_heads * c.vit_hd_pad
            mm = c.merge ** 2
            self._vbufs = dict(
                patches=torch.zeros(rows, c.kpad, dtype=bf, device=dev), x=torch.zeros(rows, D, dtype=bf, device=dev),
                xn=torch.empty(rows, D, dtype=bf, device=dev), qkv=torch.empty(rows, 3 * DH, dtype=bf, device=dev),
                q=torch.empty(rows * DH, dtype=bf, device=dev), k=torch.empty(rows * DH, dtype=bf, device=dev),
                vt=torch.zeros(rows * DH + 64, dtype=bf, device=dev), attn=torch.empty(rows, DH, dtype=bf, device=dev),
                mlp=torch.empty(rows, c.mlp_dim, dtype=bf, device=dev),
                merge_mid=torch.empty(rows // mm, D * mm, dtype=bf, device=dev))
            if self.fp8:
                self._vbufs["q8"] = torch.empty(rows * max(D, DH, c.mlp_dim), dtype=torch.uint8, device=dev)
                self._vbufs["q8s"] = torch.empty(rows, dtype=torch.float32, device=dev)
            self._vit_rows = rows
            self._vws = _lib.VitWs(**{k: _lib.ptr(v) for k, v in self._vbufs.items()})
            self._vit_layout = None
        return self._vws

    def _check_status(self) -> None:
        """After a synchronisation: did a decode step meet a read outside its invariants (position before / past the rope
        table, cache slot outside the cache)?  The kernels skip such a read instead of repairing it; here it becomes an error."""
        bits = int(self.status.item())
        if bits:
            self.status.zero_()
            raise _lib.HwocrError(f"decode step reported status {bits:#x} (HWOCR_STATUS_BAD_POSITION: a read's lens / rope_delta "
                                  "left the cache or the rope table); the reads of this call are invalid")

    def _remember_graph(self, key, g) -> None:
        """Keep a captured decode graph under `key` (= everything baked into its kernel arguments); least recently used out."""
        while len(self._graphs) >= self.MAX_GRAPHS:
            self.lib.hwocr_decode_graph_destroy(self._graphs.pop(next(iter(self._graphs))))
        self._graphs[key] = g

    def _use_graph(self, key):
        g = self._graphs.pop(key)
        self._graphs[key] = g  # most recently used last
        return g

    def _drop_graphs(self) -> None:
        for g in self._graphs.values():
            self.lib.hwocr_decode_graph_destroy(g)
        self._graphs = {}

    def close(self) -> None:
        self._drop_graphs()

    def lane(self) -> "ReadEngine":
        """A second set of read slots over the SAME weights: own KV cache, generation state, workspaces and captured graphs; the
        bound weight tensors and the model structs (read-only at run time) are shared.  Two lanes driven from two host threads on
        two HIP streams let one batch's tower + prefill (matrix pipes) run beside another batch's decode (HBM): pipeline.LanePipeline."""
        import copy

        other = copy.copy(self)
        other._graphs, other.timings, other._keep_tmp = {}, {}, None
        other._alloc_state()
        other._vit_layout = None
        return other

    # ------------------------------------------------------------------------------------------ vision tower
    def encode_pages(self, pages: list[np.ndarray], shapes: list | None = None) -> tuple[torch.Tensor, list[tuple[int, int, int]], list[np.ndarray]]:
        """uint8 [H, W, 3] pages already at tower resolution -> (embedding buffer [rows][hidden], grids, and for each
        page the buffer row of every image token in prompt (raster) order).  Pages of equal size are batched
        `vit_batch` at a time.  The Qwen2.5-VL tower works in window order from the patch gather on; its rows are never
        moved back — the row table is the library's `merged[argsort(window_index)]` (HF modeling_qwen2_5_vl.py:474-476).
        `shapes` (optional, (H, W) per page): `pages` is then indexed only when a launch group needs its pages — a lazy sequence
        (batch._LazyDeviceReads) can still be decoding the later pages while the first groups' tower launches are queued."""
        c = self.cfg
        st = _lib.stream_handle()
        mm = c.merge ** 2
        v25 = c.family == "qwen2_5_vl"
        shp = [(int(s[0]), int(s[1])) for s in (shapes if shapes is not None else [p.shape for p in pages])]
        grids = [(1, h // c.patch_size, w // c.patch_size) for h, w in shp]
        tok_rows: list = [None] * len(shp)
        chunks = []
        total = 0
        by_shape: dict = {}
        for i, hw_ in enumerate(shp):
            by_shape.setdefault(hw_, []).append(i)
        for (H, W), idxs in by_shape.items():
            gh, gw = H // c.patch_size, W // c.patch_size
            if H % (c.patch_size * c.merge) or W % (c.patch_size * c.merge) or gh < 1 or gw < 1:
                raise ValueError(f"page of {H}x{W} pixels is not a whole number of {c.patch_size * c.merge}-pixel merge blocks")
            if c.family != "paligemma" and max(gh, gw) > self.VIT_MAX_GRID:
                raise ValueError(f"page grid {gh}x{gw} exceeds the vision rotary table ({self.VIT_MAX_GRID} positions per axis)")
            P = gh * gw
            Pp = _ceil(P, 64)
            ph, pw = imageproc.vision_positions(gh, gw, c.merge)
            if c.family == "paligemma":  # no rotary in SigLIP: position 0 is the identity row of the table
                ph, pw = np.zeros_like(ph), np.zeros_like(pw)
            slot = np.arange(P // mm, dtype=np.int32)  # buffer slot (within the page) of merged token i
            if v25:
                order, win_lens = imageproc.window_order(gh, gw, c.merge, c.window_size, c.patch_size)
                row_src = (order[:, None] * mm + np.arange(mm, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
                ph, pw = ph[row_src], pw[row_src]
                slot = np.argsort(order).astype(np.int32)
                win_start = np.concatenate([[0], np.cumsum(win_lens)[:-1]]).astype(np.int32)
            for s in range(0, len(idxs), self.vit_batch):
                group = idxs[s: s + self.vit_batch]
                n = len(group)
                rows = n * Pp
                ws = self._vit_ws(_ceil(self.vit_batch * Pp, 64))
                layout = (n, Pp, gh, gw)
                if self._vit_layout != layout:
                    # per-layout index tables; pad rows keep zero patches and position 0
                    hh = np.zeros(rows, np.int32)
                    wwp = np.zeros(rows, np.int32)
                    for j in range(n):
                        hh[j * Pp: j * Pp + P], wwp[j * Pp: j * Pp + P] = ph, pw
                    t = dict(pos_h=torch.from_numpy(hh).to(self.dev), pos_w=torch.from_numpy(wwp).to(self.dev),
                             seg=torch.full((n,), P, dtype=torch.int32, device=self.dev))
                    lay = _lib.VitLayout(pos_h=_lib.ptr(t["pos_h"]), pos_w=_lib.ptr(t["pos_w"]), seg_lens=_lib.ptr(t["seg"]))
                    if v25:
                        off = np.concatenate([win_start + j * Pp for j in range(n)]).astype(np.int32)
                        t["row_src"] = torch.from_numpy(row_src).to(self.dev)
                        t["win_off"] = torch.from_numpy(off).to(self.dev)
                        t["win_lens"] = torch.from_numpy(np.tile(win_lens, n)).to(self.dev)
                        lay.row_src, lay.win_off, lay.win_lens = (_lib.ptr(t[k]) for k in ("row_src", "win_off", "win_lens"))
                        lay.nwin, lay.max_win = len(off), int(win_lens.max())
                    self._vit_tables, self._vit_lay = t, lay
                    self._vbufs["patches"].zero_()
                    self._vit_layout = layout
                items = [pages[i] for i in group]
                if any(tuple(int(v) for v in it.shape[:2]) != (H, W) for it in items):
                    raise ValueError(f"a page of this launch group is not {H}x{W} pixels as announced")
                if any(isinstance(it, torch.Tensor) for it in items):  # (some) already resident in HBM
                    imgs = torch.stack([it if isinstance(it, torch.Tensor) else
                                        torch.from_numpy(np.ascontiguousarray(it)).to(self.dev) for it in items]).contiguous()
                else:
                    imgs = torch.from_numpy(np.stack(items)).to(self.dev, non_blocking=True)
                out = torch.empty(rows // mm, c.hidden, dtype=torch.bfloat16, device=self.dev)
                _lib.check(self.lib.hwocr_vit_forward(C.byref(self.vit), C.byref(ws), _lib.ptr(imgs), n, H, W, Pp,
                                                      C.byref(self._vit_lay), _lib.ptr(out), st), "hwocr_vit_forward")
                self._keep_tmp = imgs
                for j, i in enumerate(group):
                    tok_rows[i] = total + j * (Pp // mm) + slot
                total += rows // mm
                chunks.append(out)
        emb = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=0)
        return emb, grids, tok_rows

    # ------------------------------------------------------------------------------------------ generate
    def _sampling(self, sample: dict | None) -> tuple:
        """(do_sample, temperature, top_k, top_p, seed) of a call: the checkpoint's generation_config unless `sample` overrides it."""
        c = self.cfg
        if sample is None:
            if not c.do_sample:
                return (0, 1.0, 0, 1.0, 0)
            t, k, p, seed = c.temperature, c.top_k, c.top_p, self.seed
        elif not sample:
            return (0, 1.0, 0, 1.0, 0)
        else:
            t, k, p = float(sample.get("temperature", 1.0)), int(sample.get("top_k", 0)), float(sample.get("top_p", 1.0))
            seed = int(sample.get("seed", self.seed)) & (2 ** 64 - 1)
        if not t > 0 or not p > 0:
            raise ValueError(f"sampling needs temperature > 0 and top_p > 0 (got {t}, {p})")
        if k == 1:
            return (0, 1.0, 0, 1.0, 0)  # the argmax, whatever is drawn
        return (1, t, k if 0 < k < c.vocab else 0, min(p, 1.0), seed)

    def generate(self, pages: list[np.ndarray], prompts: list[np.ndarray], max_new: int, min_new: int = 0,
                 forced: np.ndarray | None = None, return_logits: bool = False, use_graph: bool = True,
                 repetition_penalty: float | None = None, sample: dict | None = None, read_base: int = 0, hooks=None):
        """Reads: greedy, or drawn (cfg.do_sample from generation_config.json, or `sample` = dict(temperature, top_k, top_p[, seed])
        for this call; `sample={}` forces greedy): read i draws from the RNG stream of read number read_base + i, whichever slot or
        batch it is decoded in.  pages[i]: uint8 [H, W, 3] at tower resolution; prompts[i]: int32 token ids containing one run
        of image placeholders sized for pages[i].  Returns list of generated-token lists (and, for tests, the per-step
        logits of every read when return_logits; `forced` [R][max_new] teacher-forces the fed tokens)."""
        c, lib, dev = self.cfg, self.lib, self.dev
        R = len(pages)
        if R == 0:
            return []
        if R > self.max_reads:
            raise ValueError(f"{R} reads exceed max_reads={self.max_reads}")
        st = _lib.stream_handle()
        marks = []

        def mark(name):
            if self.collect_timings:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))

        if hooks is not None:   # pipeline.LanePipeline: this batch's tower waits (on the device) for the previous batch's prefill
            hooks.tower_begin()
        mark("start")
        emb, grids, tok_rows = self.encode_pages(pages)
        mark("vision")
        T = [len(p) for p in prompts]
        Tp = _ceil(max(T), 64)
        if Tp + max_new > self.ctx:
            raise ValueError(f"prompt ({max(T)}) + max_new ({max_new}) exceeds the KV cache length {self.ctx}")
        ids = np.zeros((R, Tp), np.int32)
        img_row = np.full((R, Tp), -1, np.int32)
        pos3 = np.zeros((3, R, Tp), np.int32)
        delta = np.zeros(R, np.int32)
        for r in range(R):
            p = np.asarray(prompts[r], np.int32)
            ids[r, : T[r]] = p
            m = np.nonzero(p == c.image_token_id)[0]
            n_img = grids[r][1] * grids[r][2] // c.merge ** 2
            if len(m) != n_img:
                raise ValueError(f"read {r}: {len(m)} image placeholders but the page yields {n_img} image tokens")
            img_row[r, m] = tok_rows[r]
            if c.family == "paligemma":  # 1-indexed plain positions (HF paligemma/modeling_paligemma.py:237)
                pos3[:, r, : T[r]], delta[r] = np.arange(1, T[r] + 1, dtype=np.int32), 1
            else:
                pos3[:, r, : T[r]], delta[r] = imageproc.mrope_positions(p, c.image_token_id, [grids[r]], c.merge)
        if int(pos3.max()) + max_new + 2 > self.max_pos:
            raise ValueError("rope table too short for this prompt")
        d_ids = torch.from_numpy(ids).to(dev)
        d_img = torch.from_numpy(img_row).to(dev)
        d_pos = torch.from_numpy(pos3).to(dev)
        lens_h = torch.tensor(T, dtype=torch.int32)
        d_seq = lens_h.to(dev)
        self.lens[:R].copy_(d_seq)
        self.rope_delta[:R].copy_(torch.from_numpy(delta).to(dev))
        self.n_gen[:R].zero_()
        self.finished[:R].zero_()
        if max_new not in self._tok_bufs:  # stable address per max_new: captured graphs write into it
            self._tok_bufs[max_new] = torch.empty((self.max_reads, max_new), dtype=torch.int32, device=dev)
        self.out_tokens = self._tok_bufs[max_new]
        self.out_tokens.fill_(c.pad_id)
        eos = (C.c_int * 4)(*(list(c.eos_ids) + [0] * 4)[:4])
        # repetition penalty (HF RepetitionPenaltyLogitsProcessor): a bitmap of the ids in prompt + output per read
        rp = float(c.repetition_penalty if repetition_penalty is None else repetition_penalty)
        seen_ld = (c.vocab + 31) // 32
        if rp != 1.0:
            if self._seen is None:
                self._seen = torch.zeros(self.max_reads, seen_ld, dtype=torch.int32, device=dev)  # stable address (graphs)
            bits = np.zeros((R, seen_ld * 32), np.uint8)
            for r in range(R):
                bits[r, np.asarray(prompts[r], np.int64)] = 1
            words = np.packbits(bits, axis=1, bitorder="little").view(np.uint32).astype(np.int64).astype(np.int32, casting="unsafe")
            self._seen[:R].copy_(torch.from_numpy(words.reshape(R, seen_ld)).to(dev))
        smp = self._sampling(sample)
        if smp[0]:
            self.read_ids[:R].copy_(torch.arange(read_base, read_base + R, dtype=torch.int32))
        gs = _lib.GenState(cur_ids=_lib.ptr(self.cur_ids), lens=_lib.ptr(self.lens), n_gen=_lib.ptr(self.n_gen),
                           finished=_lib.ptr(self.finished), out_tokens=_lib.ptr(self.out_tokens),
                           rope_delta=_lib.ptr(self.rope_delta), max_new=max_new, min_new=min_new,
                           n_eos=min(len(c.eos_ids), 4), pad_id=c.pad_id, eos=eos,
                           seen=_lib.ptr(self._seen) if rp != 1.0 else None, seen_ld=seen_ld, rep_penalty=rp,
                           status=_lib.ptr(self.status), do_sample=smp[0], temperature=smp[1], top_k=smp[2], top_p=smp[3],
                           seed=smp[4], read_ids=_lib.ptr(self.read_ids))
        pb = min(self.prefill_batch, R)
        ws = self._dec_ws(max(pb * Tp, self.max_reads))
        step_logits = [] if return_logits else None
        first_logits = []
        chunk_keep = []
        for s0 in range(0, R, pb):
            n = min(pb, R - s0)
            last = torch.tensor([j * Tp + T[s0 + j] - 1 for j in range(n)], dtype=torch.int32, device=dev)
            pos_chunk = d_pos[:, s0: s0 + n].contiguous()
            _lib.check(lib.hwocr_prefill(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs),
                                         _lib.ptr(d_ids[s0: s0 + n]), _lib.ptr(d_img[s0: s0 + n]), _lib.ptr(emb),
                                         _lib.ptr(pos_chunk), _lib.ptr(d_seq[s0: s0 + n]), _lib.ptr(last), n, Tp, s0,
                                         max(T[s0: s0 + n]), st), "hwocr_prefill")
            if return_logits:
                first_logits.append(self._bufs["logits"][:n].clone())
            chunk_keep.append((last, pos_chunk))  # index tensors stay alive until the stream has consumed them (no host sync per chunk)
        mark("prefill")
        if hooks is not None:   # ... and its decode for the previous batch's decode
            hooks.prefill_end()
            st = _lib.stream_handle()  # (a "partition" pipeline moves the decode onto its own CU-masked stream here)
        if return_logits:
            step_logits.append(torch.cat(first_logits, dim=0))
        def feed(col):  # teacher forcing: the fed token replaces the chosen one (the select kernel adds what it was fed
            self.cur_ids[:R].copy_(torch.from_numpy(np.ascontiguousarray(forced[:, col]).astype(np.int32)).to(dev))  # to the bitmap)

        if forced is not None:
            feed(0)
        splits = self.attn_splits or pick_attn_splits(R, c.kv_heads)
        steps = max_new - 1
        if use_graph and not return_logits and forced is None and steps > 0:
            key = (R, splits, max_new, min_new, rp, tuple(c.eos_ids), c.pad_id, smp)  # everything the captured launches carry by value
            if key not in self._graphs:
                # one eager step first: lazy one-time kernel attributes must not be set inside a capture
                _lib.check(lib.hwocr_decode_step(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs), R, splits, st))
                steps -= 1
                torch.cuda.current_stream().synchronize()
                g = C.c_void_p()
                self._gs_keep = (gs, eos)
                _lib.check(lib.hwocr_decode_graph_create(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs), R,
                                                         splits, C.byref(g)), "hwocr_decode_graph_create")
                self._remember_graph(key, g)
            graph = self._use_graph(key)
            done = 0
            while done < steps:
                n = min(32, steps - done)
                _lib.check(lib.hwocr_decode_graph_launch(graph, n, st), "hwocr_decode_graph_launch")
                done += n
                if min_new < max_new and bool(self.finished[:R].all()):
                    break
        else:
            for i in range(steps):
                _lib.check(lib.hwocr_decode_step(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs), R, splits, st))
                if return_logits:
                    step_logits.append(self._bufs["logits"][:R].clone())
                if forced is not None and i + 1 < max_new:
                    feed(i + 1)
        mark("decode")
        if hooks is not None:
            hooks.decode_end()
        torch.cuda.current_stream().synchronize()
        self._check_status()
        if marks:
            self.timings = {marks[i][0] + "_ms": marks[i - 1][1].elapsed_time(marks[i][1]) for i in range(1, len(marks))}
            self.timings["decode_steps"] = max_new - 1
        toks = self.out_tokens[:R].cpu().numpy()
        ng = self.n_gen[:R].cpu().numpy()
        out = []
        for r in range(R):
            seq = toks[r, : min(int(ng[r]), max_new)].tolist()
            # drop what follows the first EOS (finished reads keep emitting pad in lockstep)
            for k, t in enumerate(seq):
                if t in c.eos_ids and k + 1 >= min_new:
                    seq = seq[: k + 1]
                    break
            out.append(seq)
        if return_logits:
            return out, torch.stack(step_logits, dim=1)  # [R][steps][V]
        return out

    # ------------------------------------------------------------------------------------------ continuous batching
    def generate_stream(self, pages: list, prompts: list, max_new: int, min_new: int = 0, sync_every: int = 16,
                        repetition_penalty: float | None = None, min_admit: int = 0, sample: dict | None = None,
                        read_ids: list | None = None, on_done=None, source: "ReadSource | None" = None,
                        max_slots: int = 0) -> list[list[int]]:
        """Reads (greedy or drawn, as `generate`; read i draws from the RNG stream of read number read_ids[i], default i) of ANY number of (page, prompt) pairs through the engine's `max_reads` decode slots, refilled as
        reads finish (EOS or max_new): the lockstep `generate` keeps a whole batch decoding until its longest read is done,
        which with the reference's 2048-token budget (config.py:19) and pages of a few hundred tokens idles most slots.
        Every `sync_every` decode steps the host looks at the stop flags, harvests finished reads, and prefills new ones
        into the freed slots (vision + prefill run while the other slots wait; their KV stays in place) — once at least
        `min_admit` slots are free (default an eighth of the slots: the tower and the prefill GEMMs want rows), or nothing
        is decoding.  Results are in input order and equal `generate` on each read alone (reads never see each other).
        `on_done(i, tokens)` is called (on this thread) as soon as read i is harvested: the batch driver hands finished pages to its
        gather / writer threads while the remaining reads decode.  `source`: a ReadSource over the same (pages, prompts) shared with
        other lanes' generate_stream calls — this call then reads what it takes from it, the result list holds None for reads another
        lane took.  `max_slots`: use at most that many of the engine's decode slots (tools.plan_lanes balances a job over lanes)."""
        c, lib, dev = self.cfg, self.lib, self.dev
        N = len(pages)
        if N == 0:
            return []
        R = min(self.max_reads, N, max_slots or self.max_reads)
        src = source if source is not None else ReadSource(N)
        st = _lib.stream_handle()
        Tmax = max(len(p) for p in prompts)
        if _ceil(Tmax, 64) + max_new + sync_every > self.ctx:
            raise ValueError(f"prompt ({Tmax}) + max_new ({max_new}) + sync_every ({sync_every}) exceeds the KV cache length {self.ctx}")
        if max_new not in self._tok_bufs:
            self._tok_bufs[max_new] = torch.empty((self.max_reads, max_new), dtype=torch.int32, device=dev)
        self.out_tokens = self._tok_bufs[max_new]
        eos = (C.c_int * 4)(*(list(c.eos_ids) + [0] * 4)[:4])
        rp = float(c.repetition_penalty if repetition_penalty is None else repetition_penalty)
        seen_ld = (c.vocab + 31) // 32
        if rp != 1.0 and self._seen is None:
            self._seen = torch.zeros(self.max_reads, seen_ld, dtype=torch.int32, device=dev)
        smp = self._sampling(sample)
        gs = _lib.GenState(cur_ids=_lib.ptr(self.cur_ids), lens=_lib.ptr(self.lens), n_gen=_lib.ptr(self.n_gen),
                           finished=_lib.ptr(self.finished), out_tokens=_lib.ptr(self.out_tokens),
                           rope_delta=_lib.ptr(self.rope_delta), max_new=max_new, min_new=min_new,
                           n_eos=min(len(c.eos_ids), 4), pad_id=c.pad_id, eos=eos,
                           seen=_lib.ptr(self._seen) if rp != 1.0 else None, seen_ld=seen_ld, rep_penalty=rp,
                           status=_lib.ptr(self.status), do_sample=smp[0], temperature=smp[1], top_k=smp[2], top_p=smp[3],
                           seed=smp[4], read_ids=_lib.ptr(self.read_ids))
        # idle slots decode a finished one-token read (cheap) until a real read moves in
        self.finished[:R].fill_(1)
        self.lens[:R].fill_(1)
        self.rope_delta[:R].zero_()   # a parked slot's position is lens - 1 + delta: a stale negative delta would index before the rope table
        self.n_gen[:R].zero_()
        self.cur_ids[:R].fill_(c.pad_id)
        slot_read = [-1] * R
        results: list = [None] * N
        splits = self.attn_splits or pick_attn_splits(R, c.kv_heads)
        ws = self._dec_ws(max(min(self.prefill_batch, R) * _ceil(Tmax, 64), self.max_reads))

        admit_keep: list = []  # device tensors the queued prefill launches read; released after the next synchronisation

        def admit(slots: list[int]) -> int:
            reads = src.take(len(slots))
            if not reads:
                return 0
            slots = slots[: len(reads)]
            if hasattr(pages, "shape_of"):   # a lazy sequence: its items are made when the tower's launch groups reach them
                emb, grids, tok_rows = self.encode_pages(_IndexView(pages, reads), shapes=[pages.shape_of(r) for r in reads])
            else:
                emb, grids, tok_rows = self.encode_pages([pages[r] for r in reads])
            T = [len(prompts[r]) for r in reads]
            Tp = _ceil(max(T), 64)
            n = len(reads)
            ids = np.zeros((n, Tp), np.int32)
            img_row = np.full((n, Tp), -1, np.int32)
            pos3 = np.zeros((3, n, Tp), np.int32)
            delta = np.zeros(n, np.int32)
            for j, r in enumerate(reads):
                p = np.asarray(prompts[r], np.int32)
                ids[j, : T[j]] = p
                m = np.nonzero(p == c.image_token_id)[0]
                if len(m) != len(tok_rows[j]):
                    raise ValueError(f"read {r}: {len(m)} image placeholders but the page yields {len(tok_rows[j])} image tokens")
                img_row[j, m] = tok_rows[j]
                if c.family == "paligemma":
                    pos3[:, j, : T[j]], delta[j] = np.arange(1, T[j] + 1, dtype=np.int32), 1
                else:
                    pos3[:, j, : T[j]], delta[j] = imageproc.mrope_positions(p, c.image_token_id, [grids[j]], c.merge)
            d_ids, d_img = torch.from_numpy(ids).to(dev), torch.from_numpy(img_row).to(dev)
            d_pos, d_seq = torch.from_numpy(pos3).to(dev), torch.tensor(T, dtype=torch.int32, device=dev)
            sl = torch.tensor(slots, dtype=torch.long, device=dev)
            admit_keep.append((d_ids, d_img, d_pos, d_seq, emb))  # alive until the next host sync of the decode loop
            self.lens[sl] = d_seq
            self.read_ids[sl] = torch.tensor(reads if read_ids is None else [int(read_ids[r]) for r in reads], dtype=torch.int32, device=dev)
            self.rope_delta[sl] = torch.from_numpy(delta).to(dev)
            self.n_gen[sl] = 0
            self.finished[sl] = 0
            self.out_tokens[sl] = c.pad_id
            if rp != 1.0:
                bits = np.zeros((n, seen_ld * 32), np.uint8)
                for j, r in enumerate(reads):
                    bits[j, np.asarray(prompts[r], np.int64)] = 1
                words = np.packbits(bits, axis=1, bitorder="little").view(np.uint32).astype(np.int64).astype(np.int32, casting="unsafe")
                self._seen[sl] = torch.from_numpy(words.reshape(n, seen_ld)).to(dev)
            # prefill writes consecutive cache slots: one call per run of adjacent slots (slots are sorted)
            j0 = 0
            while j0 < n:
                j1 = j0 + 1
                while j1 < n and slots[j1] == slots[j1 - 1] + 1 and j1 - j0 < self.prefill_batch:
                    j1 += 1
                k = j1 - j0
                last = torch.tensor([i * Tp + T[j0 + i] - 1 for i in range(k)], dtype=torch.int32, device=dev)
                pos_chunk = d_pos[:, j0:j1].contiguous()
                _lib.check(lib.hwocr_prefill(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs),
                                             _lib.ptr(d_ids[j0:j1]), _lib.ptr(d_img[j0:j1]), _lib.ptr(emb), _lib.ptr(pos_chunk),
                                             _lib.ptr(d_seq[j0:j1]), _lib.ptr(last), k, Tp, slots[j0], max(T[j0:j1]), st),
                           "hwocr_prefill")
                admit_keep.append((last, pos_chunk))
                j0 = j1
            for s, r in zip(slots, reads):
                slot_read[s] = r
            return len(reads)

        key = (R, splits, max_new, min_new, rp, tuple(c.eos_ids), c.pad_id, smp)
        trace = [] if self.collect_timings else None   # (reads admitted, wall ms) per trip of the loop: where a folder's time goes
        while True:
            t_trip = time.perf_counter()
            n_admit = 0
            free = [s for s in range(R) if slot_read[s] < 0]
            left = src.remaining()
            if free and left > 0 and (len(free) >= (min_admit or max(1, R // 8)) or len(free) == R or left <= len(free)):
                n_admit = admit(free[:left])
            if all(r < 0 for r in slot_read):
                break
            if key not in self._graphs:
                _lib.check(lib.hwocr_decode_step(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs), R, splits, st))
                torch.cuda.current_stream().synchronize()
                g = C.c_void_p()
                self._gs_keep = (gs, eos)
                _lib.check(lib.hwocr_decode_graph_create(C.byref(self.dec), C.byref(ws), C.byref(self.kv), C.byref(gs), R, splits,
                                                         C.byref(g)), "hwocr_decode_graph_create")
                self._remember_graph(key, g)
                _lib.check(lib.hwocr_decode_graph_launch(g, sync_every - 1, st), "hwocr_decode_graph_launch")
            else:
                _lib.check(lib.hwocr_decode_graph_launch(self._use_graph(key), sync_every, st), "hwocr_decode_graph_launch")
            torch.cuda.current_stream().synchronize()
            admit_keep.clear()
            self._check_status()
            fin = self.finished[:R].cpu().numpy()
            ng = self.n_gen[:R].cpu().numpy()
            done = [s for s in range(R) if slot_read[s] >= 0 and (fin[s] or ng[s] >= max_new)]
            if done:
                toks = self.out_tokens[:R].cpu().numpy()
                for s in done:
                    seq = toks[s, : min(int(ng[s]), max_new)].tolist()
                    for k, t in enumerate(seq):
                        if t in c.eos_ids and k + 1 >= min_new:
                            seq = seq[: k + 1]
                            break
                    results[slot_read[s]] = seq
                    if on_done is not None:
                        on_done(slot_read[s], seq)
                    slot_read[s] = -1
            idle = torch.tensor([s for s in range(R) if slot_read[s] < 0], dtype=torch.long, device=dev)
            if len(idle):  # parked: finished one-token reads at position 0 (their context must not grow with the padding steps)
                self.finished[idle] = 1
                self.lens[idle] = 1
                self.rope_delta[idle] = 0
            if trace is not None:
                trace.append((n_admit, (time.perf_counter() - t_trip) * 1e3))
        if trace is not None:
            self.stream_trace = trace
        return results
