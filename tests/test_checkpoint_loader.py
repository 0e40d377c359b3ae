"""`engine.load_checkpoint_dir` on checkpoint directories written by the real HF classes (config.json + safetensors of the
tiny golden models, all three families): family detection, every ModelConfig field the kernels depend on, key
normalisation (current and pre-4.52 parameter prefixes).  CPU only — the loader is host code."""
import os

import pytest
import torch
from safetensors.torch import load_file, save_file

from handwritten_ocr_amd import engine
from tests._golden import GOLD, load_json


def _write_dir(tmp_path, family):
    import transformers as tf

    if family == "paligemma":
        m = load_json("paligemma_tiny.json")["config"]
        cfg = tf.PaliGemmaConfig(vision_config=dict(m["vision"]), text_config=dict(m["text"]), image_token_id=m["image_token_id"],
                                 projection_dim=256, hidden_size=256, vocab_size=512, pad_token_id=m["pad"], bos_token_id=2,
                                 eos_token_id=m["eos"])
        weights = "paligemma_tiny_weights.safetensors"
    else:
        stem = {"qwen2_vl": "qwen2vl_tiny", "qwen2_5_vl": "qwen25vl_tiny"}[family]
        m = load_json(stem + ".json")["config"]
        cls = tf.Qwen2VLConfig if family == "qwen2_vl" else tf.Qwen2_5_VLConfig
        cfg = cls(vision_config=dict(m["vision"]), text_config=dict(m["text"]), image_token_id=m["image_token_id"],
                  video_token_id=m["video_token_id"], vision_start_token_id=m["vision_start_token_id"],
                  vision_end_token_id=m["vision_end_token_id"], tie_word_embeddings=True)
        weights = stem + "_weights.safetensors"
    cfg.save_pretrained(tmp_path)
    sd = load_file(os.path.join(GOLD, weights))
    save_file(sd, os.path.join(tmp_path, "model.safetensors"))
    return sd


@pytest.mark.parametrize("family,preset", [("qwen2_vl", "tiny"), ("qwen2_5_vl", "tiny25"), ("paligemma", "tinypg")])
def test_loader_reads_hf_checkpoint_dir(tmp_path, family, preset):
    sd = _write_dir(tmp_path, family)
    cfg, loaded = engine.load_checkpoint_dir(str(tmp_path), device="cpu")
    want = engine.preset(preset)
    cfg.validate()
    assert cfg.family == family
    for f in ("depth", "embed_dim", "num_heads", "patch_size", "merge", "tps", "hidden", "layers", "q_heads", "kv_heads",
              "inter", "vocab", "head_dim", "rope_theta", "image_token_id", "mlp_dim", "vit_hd", "vit_hd_pad", "kpad"):
        assert getattr(cfg, f) == getattr(want, f), f
    if family == "qwen2_5_vl":
        assert (cfg.vit_inter, cfg.window_size, tuple(cfg.fullatt)) == (want.vit_inter, want.window_size, tuple(want.fullatt))
    if family == "paligemma":
        assert (cfg.vit_inter, cfg.image_size, cfg.bos_id) == (want.vit_inter, want.image_size, want.bos_id)
    assert set(loaded) == set(sd) and all(v.dtype == torch.bfloat16 for v in loaded.values())


def test_key_normalisation_of_older_layouts():
    old = {"visual.blocks.0.norm1.weight": 1, "model.layers.0.mlp.up_proj.weight": 2, "model.norm.weight": 3,
           "lm_head.weight": 4, "model.visual.merger.ln_q.weight": 5,
           "vision_tower.vision_model.encoder.layers.0.layer_norm1.weight": 6, "multi_modal_projector.linear.bias": 7,
           "language_model.model.layers.1.self_attn.q_proj.weight": 8,
           "model.vision_tower.vision_model.post_layernorm.bias": 9}
    new = engine.normalize_keys(old)
    assert new == {"model.visual.blocks.0.norm1.weight": 1, "model.language_model.layers.0.mlp.up_proj.weight": 2,
                   "model.language_model.norm.weight": 3, "lm_head.weight": 4, "model.visual.merger.ln_q.weight": 5,
                   "model.vision_tower.encoder.layers.0.layer_norm1.weight": 6, "model.multi_modal_projector.linear.bias": 7,
                   "model.language_model.layers.1.self_attn.q_proj.weight": 8, "model.vision_tower.post_layernorm.bias": 9}


@pytest.mark.parametrize("gen,eos,pad,rp,note,smp", [
    ({"do_sample": True, "temperature": 0.01, "top_k": 1, "top_p": 0.001, "repetition_penalty": 1.05,
      "eos_token_id": [505, 510], "pad_token_id": 511}, (505, 510), 511, 1.05, "argmax", (False, 1.0, 0, 1.0)),  # the Qwen2-VL model-card shape
    ({"do_sample": True, "temperature": 0.8, "top_k": 50, "eos_token_id": 505}, (505,), 511, 1.0, "hwocr_sample_advance", (True, 0.8, 50, 1.0)),
    ({"do_sample": True, "top_p": 0.9, "eos_token_id": 505}, (505,), 511, 1.0, "hwocr_sample_advance", (True, 1.0, 50, 0.9)),  # HF defaults: T 1, top_k 50
    ({"eos_token_id": 510}, (510,), 511, 1.0, "", (False, 1.0, 0, 1.0)),
])
def test_generation_config_defaults(tmp_path, gen, eos, pad, rp, note, smp):
    """generation_config.json written by HF's own GenerationConfig.save_pretrained: EOS set, pad id, repetition penalty and the
    sampling settings are taken over (HF's defaults where the file is silent); top_k = 1 is the argmax and stays greedy."""
    import transformers as tf

    _write_dir(tmp_path, "qwen2_vl")
    tf.GenerationConfig(**gen).save_pretrained(tmp_path)
    cfg, _ = engine.load_checkpoint_dir(str(tmp_path), device="cpu")
    assert (tuple(cfg.eos_ids), cfg.repetition_penalty) == (eos, rp)
    assert cfg.pad_id == (gen.get("pad_token_id", engine.ModelConfig().pad_id))
    assert note in cfg.sampling_note
    assert (cfg.do_sample, cfg.temperature, cfg.top_k, cfg.top_p) == smp
