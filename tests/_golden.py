"""Loading helpers for tests/golden (data only: inputs + expected outputs written by tools/make_goldens.py)."""
import json
import os

import torch
from safetensors.torch import load_file

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLD, name), "r", encoding="utf-8") as f:
        return json.load(f)


STEM = {"qwen2_vl": "qwen2vl_tiny", "qwen2_5_vl": "qwen25vl_tiny"}
FAMILIES = tuple(STEM)


def tiny_meta(family="qwen2_vl"):
    return load_json(STEM[family] + ".json")


def tiny_weights(dtype=torch.bfloat16, family="qwen2_vl"):
    sd = load_file(os.path.join(GOLD, STEM[family] + "_weights.safetensors"))
    return {k: v.to(dtype) for k, v in sd.items()}


def tiny_case(tag, family="qwen2_vl"):
    return load_file(os.path.join(GOLD, f"{STEM[family]}_{tag}.safetensors"))


def tiny_ref_config(family="qwen2_vl"):
    from oracle.qwen2vl_ref import RefConfig

    m = tiny_meta(family)["config"]
    v, t = m["vision"], m["text"]
    if family == "qwen2_vl":
        tower = dict(embed_dim=v["embed_dim"], mlp_ratio=v["mlp_ratio"])
    else:
        tower = dict(embed_dim=v["hidden_size"], family=family, vit_inter=v["intermediate_size"],
                     window_size=v["window_size"], fullatt=tuple(v["fullatt_block_indexes"]))
    return RefConfig(depth=v["depth"], num_heads=v["num_heads"], patch_size=v["patch_size"], merge=v["spatial_merge_size"],
                     tps=v["temporal_patch_size"], hidden=t["hidden_size"], layers=t["num_hidden_layers"],
                     q_heads=t["num_attention_heads"], kv_heads=t["num_key_value_heads"], inter=t["intermediate_size"],
                     vocab=t["vocab_size"], rope_theta=t["rope_parameters"]["rope_theta"],
                     mrope_section=tuple(t["rope_parameters"]["mrope_section"]), eps=t["rms_norm_eps"],
                     image_token_id=m["image_token_id"], vision_start_id=m["vision_start_token_id"],
                     vision_end_id=m["vision_end_token_id"], tie=True, eos_ids=(m["eos"],), pad_id=m["pad"], **tower)


# ------------------------------------------------------------------------------------------------ trained tiny checkpoints
TRAINED = {"qwen2_vl": "trained_qwen2vl", "qwen2_5_vl": "trained_qwen25vl", "paligemma": "trained_paligemma"}
TRAINED_FAMILIES = tuple(TRAINED)


def trained_meta(family="qwen2_vl"):
    """tests/golden/trained_*.json: HF's own transcriptions (token streams + decoded text) of 8 synthetic pages by the briefly
    trained tiny checkpoint in tests/golden/trained_*/ (tools/make_goldens.py --only trained,trained25)."""
    return load_json(TRAINED[family] + ".json")


def trained_dir(family="qwen2_vl"):
    return os.path.join(GOLD, TRAINED[family])


def trained_page(case):
    """The page a fixture case was transcribed from (regenerated: synth.make_page + its paper colour)."""
    from handwritten_ocr_amd.synth import make_page, tint_page

    return tint_page(make_page(case["page_seed"], *case["page_hw"]), case["page_tint"])


def mean_cer(want_texts, got_texts):
    """The reference's metric over a set of pages: cer() per page (ocr_agent/tools.py:103-118, restated bit-exactly in
    handwritten_ocr_amd.text and pinned by text_kats.json), averaged."""
    from handwritten_ocr_amd import text

    return sum(text.cer(w, g) for w, g in zip(want_texts, got_texts)) / len(want_texts)
