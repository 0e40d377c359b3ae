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
