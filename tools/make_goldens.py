#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the BUILD container, never on the GPU box).

Sources of truth
  * string / preprocessing / node control-flow KATs: the reference's own Python, imported from /root/reference with an
    inert `ollama` module in sys.modules (ocr_agent/tools.py:146 imports it at module level; nothing on the paths
    exercised here calls it).  cv2 is absent in this image, so preprocessing KATs pin the PIL fallbacks.
  * processor / chat-template / tokenizer KATs (tests/golden/tokenizer_tiny/ + tokenizer_kats.json): HF's
    PreTrainedTokenizerFast over a tiny byte-level BPE trained here, with the Qwen2-VL chat template, and HF's PIL image
    processor — the objects run_ocr's processor is made of (tools.py:744-769): rendered chat text, prompt ids, decode strings.
  * image-processor and model KATs: Hugging Face transformers (the library run_ocr drives, tools.py:690-765) on tiny
    seeded random-init Qwen2-VL models — no checkpoint is available offline.

Only data is written: inputs, expected outputs, weights of the random tiny model.  No reference source text.
Usage: python tools/make_goldens.py [--only text,preprocess,nodes,image,model,model25,paligemma,tokenizer,trained,trained25,trainedpg]
"""
from __future__ import annotations

import argparse
import hashlib
import io
import json
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference"


def _import_reference_tools():
    sys.modules.setdefault("ollama", types.ModuleType("ollama"))  # inert: never called
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import ocr_agent.tools as tools  # noqa

    return tools


# ------------------------------------------------------------------------------------------------ text
WORDS = ("the quick brown fox jumps over a lazy dog while handwritten notes drift across ruled paper and ink "
         "fades near margins where someone once wrote dates names sums and little reminders to buy milk").split()


def _rand_text(rng: random.Random, n: int) -> str:
    out = []
    for _ in range(n):
        w = rng.choice(WORDS)
        r = rng.random()
        if r < 0.08:
            w = w.capitalize()
        elif r < 0.12:
            w = w.upper()
        elif r < 0.16:
            w += rng.choice(",.;:!?")
        out.append(w)
    return " ".join(out)


def _mutate(rng: random.Random, s: str, rate: float) -> str:
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.choice("abcdefghijklmnopqrstuvwxyz "))
            continue
        out.append(ch)
        if r < rate:
            out.append(rng.choice("aeiou \n"))
    return "".join(out)


def make_text(tools) -> None:
    rng = random.Random(1234)
    pairs = [("", ""), ("", "abc"), ("abc", ""), ("kitten", "sitting"), ("a", "a"), ("A", "a"),
             ("‘quoted’ “text” – dash — em", "'quoted' \"text\" - dash - em"),
             ("  many   spaces\n\nand\tlines ", "many spaces and lines"),
             ("The Cat sat", "the cat SAT"), ("one two three", "three two one"),
             ("été naïve 中文 \U0001f600", "ete naive 中 \U0001f600"),
             ("x" * 70, "x" * 63 + "y"), ("ab" * 100, "ba" * 100)]
    for _ in range(200):
        n = rng.choice([1, 2, 5, 12, 40, 90])
        a = _rand_text(rng, n)
        b = _mutate(rng, a, rng.choice([0.0, 0.02, 0.1, 0.4]))
        pairs.append((a, b))
    cases = []
    for a, b in pairs:
        na, nb = tools.normalize_text(a), tools.normalize_text(b)
        cases.append({
            "a": a, "b": b,
            "normalize_a": na, "normalize_a_lower": tools.normalize_text(a, True),
            "levenshtein": tools.levenshtein(na, nb),
            "levenshtein_words": tools._levenshtein_words(na.split(), nb.split()),
            "cer": tools.cer(a, b), "wer": tools.wer(a, b), "cer_lower": tools.cer(a, b, True),
            "tier1": tools.tier1_metrics(a, b),
            "compare": tools.compare_versions(a, b),
        })
    merges = [[], ["solo text"], ["a b c", "a b c"], ["the Cat sat", "the cat sat"], ["The cat", "the cat", "THE cat"],
              ["a b c d", "a x c d", "a b c"], ["same len one", "same len two"], ["x", "y z", "y z w"]]
    for _ in range(80):
        base = _rand_text(rng, rng.choice([3, 10, 30, 80]))
        k = rng.choice([2, 3, 3, 4])
        merges.append([_mutate(rng, base, rng.choice([0.0, 0.03, 0.15])) for _ in range(k)])
    mcases = [{"versions": v, "merged": tools.merge_versions(v)} for v in merges]
    aligns = []
    for _ in range(40):
        a = _rand_text(rng, rng.choice([1, 4, 15, 40])).split()
        b = _mutate(rng, " ".join(a), 0.2).split()
        aligns.append({"backbone": a, "words": b, "aligned": tools._align_to_backbone(a, b)})
    long_a = _rand_text(rng, 260)
    long_b = _mutate(rng, long_a, 0.05)
    longc = {"a": long_a, "b": long_b, "compare": tools.compare_versions(long_a, long_b),
             "merged3": tools.merge_versions([long_a, long_b, _mutate(rng, long_a, 0.05)])}
    longc["third"] = None  # merged3's third input is not reproducible from the rng state; store it explicitly
    third = _mutate(random.Random(99), long_a, 0.05)
    longc["third"] = third
    longc["merged3"] = tools.merge_versions([long_a, long_b, third])
    gt_cases = []
    for text in ["# Title\n\n## Ground Truth\n\nhello world\n", "no header here\n", "## Ground Truth\n", ""]:
        import tempfile
        with tempfile.NamedTemporaryFile("w", suffix=".md", delete=False, encoding="utf-8") as f:
            f.write(text)
        gt_cases.append({"file_text": text, "parsed": tools.parse_ground_truth(f.name)})
        os.unlink(f.name)
    js = [{"raw": r, "parsed": tools.parse_json_response(r)} for r in
          ['{"a": 1}', '```json\n{"a": [1, 2]}\n```', 'noise {"k": {"n": 2}} tail', "[1, 2, 3]", "not json", "{broken"]]
    with open(os.path.join(GOLD, "text_kats.json"), "w", encoding="utf-8") as f:
        json.dump({"source": "ocr_agent.tools imported from /root/reference (inert ollama stub)", "pairs": cases,
                   "merges": mcases, "aligns": aligns, "long": longc, "ground_truth": gt_cases, "json": js}, f,
                  ensure_ascii=True, indent=0)
    print("text_kats.json:", len(cases), "pairs,", len(mcases), "merges")


# ------------------------------------------------------------------------------------------------ synthetic pages
from handwritten_ocr_amd.synth import make_page, tint_page  # noqa: E402  (same generator the bench uses)


def make_preprocess(tools) -> None:
    import tempfile
    from PIL import Image
    from ocr_agent import config

    cases = []
    tmpd = tempfile.mkdtemp()
    for seed, (h, w), mode in [(0, (64, 64), "RGB"), (1, (128, 96), "RGB"), (2, (256, 256), "RGB"), (3, (100, 140), "L")]:
        arr = make_page(seed, h, w)
        img = Image.fromarray(arr, "RGB")
        if mode == "L":
            img = img.convert("L")
        src = os.path.join(tmpd, f"page{seed}.png")
        img.save(src)
        strategies = [list(s) for s in config.PREPROCESSING_STRATEGIES] + ["original", ["original"], "sharpen",
                                                                         ["binarize", "sharpen"], ["nope", "sharpen"]]
        for strat in strategies:
            buf = io.StringIO()
            old = sys.stdout
            sys.stdout = buf
            try:
                out = tools.preprocess_image(src, strat)
            finally:
                sys.stdout = old
            same = out == src
            res = Image.open(out)
            px = np.asarray(res)
            cases.append({"seed": seed, "h": h, "w": w, "mode": mode, "strategy": strat, "returns_input_path": same,
                          "out_mode": res.mode, "out_size": list(res.size),
                          "pixels_sha256": hashlib.sha256(px.tobytes()).hexdigest(),
                          "prefix": os.path.basename(out).split("_")[0] if not same else None,
                          "basename_starts": None if same else os.path.basename(out)[: len("ocr_" + "+".join(
                              s for s in (strat if isinstance(strat, list) else [strat]) if s != "original") + "_")],
                          "suffix": os.path.splitext(out)[1], "stdout": buf.getvalue()})
            if not same:
                os.unlink(out)
    with open(os.path.join(GOLD, "preprocess_kats.json"), "w") as f:
        json.dump({"cv2_available": False, "source": "ocr_agent.tools.preprocess_image (PIL fallbacks)",
                   "strategies_config": [list(s) for s in config.PREPROCESSING_STRATEGIES], "cases": cases}, f, indent=0)
    print("preprocess_kats.json:", len(cases), "cases")


# ------------------------------------------------------------------------------------------------ nodes
def make_nodes() -> None:
    """Control-flow KAT: the reference's node_initial_ocr / node_reocr with scripted run_ocr / preprocess / arbitrator."""
    tools = _import_reference_tools()
    sys.modules.setdefault("pydantic", __import__("pydantic"))
    import ocr_agent.nodes as nodes
    from ocr_agent import config
    import time

    scripts = {
        "agree": ["the quick brown fox jumps", "the quick brown fox jumps", "never read"],
        "disagree": ["the quick brown fox jumps over", "a completely different reading here", "the quick brown fax jumps over"],
        "case_tie": ["The cat sat down", "the cat sat down", "THE cat sat down"],
    }
    out = {}
    for name, texts in scripts.items():
        calls = {"n": 0, "pre": []}

        def fake_ocr(path, params=None, _c=calls, _t=texts):
            t = _t[min(_c["n"], len(_t) - 1)]
            _c["n"] += 1
            return t

        def fake_pre(path, strategy, _c=calls):
            _c["pre"].append(strategy if isinstance(strategy, str) else list(strategy))
            return path + "#" + ("+".join(strategy) if isinstance(strategy, list) else strategy)

        nodes.run_ocr, nodes.preprocess_image, nodes.unload_ocr_model = fake_ocr, fake_pre, (lambda: None)
        state = {"image_path": "/pages/p1.png", "candidates": [], "critiques": [], "edits": [], "current_best": "",
                 "current_score": 0.0, "iteration": 0, "max_iterations": 3, "status": "running", "reason": "",
                 "strategies_used": [], "plateau_count": 0, "prev_score": 0.0, "prev_critique": None,
                 "config": {"accept_threshold": 85, "plateau_patience": 2,
                            "strategies": [list(s) for s in config.PREPROCESSING_STRATEGIES],
                            "agreement_threshold": config.AGREEMENT_THRESHOLD},
                 "trace_events": [], "start_time": time.monotonic()}
        buf = io.StringIO()
        old = sys.stdout
        sys.stdout = buf
        try:
            upd = nodes.node_initial_ocr(state)
        finally:
            sys.stdout = old

        def strip(ev):
            ev = dict(ev)
            ev.pop("timestamp"), ev.pop("elapsed_seconds")
            return ev

        rec = {"texts": texts, "update": {k: ([strip(e) for e in v] if k == "trace_events" else v) for k, v in upd.items()},
               "preprocess_calls": calls["pre"], "ocr_calls": calls["n"],
               "stdout_lines": [ln.split("] ", 1)[-1] if ln.startswith("[") else ln for ln in buf.getvalue().splitlines()]}
        # re-OCR rounds until the strategies are exhausted (arbitrator scripted: keeps the longer text)
        state.update(upd)

        class Arb:
            def __init__(self, text):
                self.final_text, self.confidence, self.decisions, self.uncertain_segments = text, 77, [], []

            def model_dump(self):
                return {"final_text": self.final_text, "confidence": self.confidence, "decisions": [], "uncertain_segments": []}

        nodes.run_arbitrator = lambda versions: Arb(max((v["text"] for v in versions), key=len))
        rounds = []
        for _ in range(6):
            sys.stdout = io.StringIO()
            try:
                u = nodes.node_reocr(state)
            finally:
                sys.stdout = old
            rounds.append({k: ([strip(e) for e in v] if k == "trace_events" else v) for k, v in u.items()})
            state.update(u)
            if u.get("reason") == "exhausted":
                break
        rec["reocr_rounds"] = rounds
        out[name] = rec
    with open(os.path.join(GOLD, "nodes_kats.json"), "w") as f:
        json.dump({"source": "ocr_agent.nodes with scripted run_ocr / preprocess_image / run_arbitrator", "cases": out}, f,
                  indent=0, default=str)
    print("nodes_kats.json:", list(out))


# ------------------------------------------------------------------------------------------------ image processor
def make_image() -> None:
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil, smart_resize
    from PIL import Image

    table = []
    for (h, w) in [(512, 512), (1024, 1024), (896, 896), (1000, 700), (333, 517), (2000, 3000), (100, 100), (28, 5000),
                   (57, 83), (4032, 3024), (255, 257), (1, 1), (27, 27), (3000, 20)]:
        for (mn, mx) in [(256 * 256, 1024 * 1024), (56 * 56, 14 * 14 * 4 * 1280)]:
            try:
                r = list(smart_resize(h, w, 28, mn, mx))
            except ValueError as e:
                r = "ValueError"
            table.append({"h": h, "w": w, "min_pixels": mn, "max_pixels": mx, "out": r})
    from safetensors.torch import save_file

    tensors = {}
    meta = []
    for idx, (seed, (h, w), (mn, mx)) in enumerate([(5, (60, 90), (28 * 28, 1024 * 1024)), (6, (150, 200), (28 * 28, 1024 * 1024)),
                                                     (7, (120, 100), (256 * 256, 1024 * 1024))]):
        arr = make_page(seed, h, w)
        proc = Qwen2VLImageProcessorPil(min_pixels=mn, max_pixels=mx)
        out = proc(images=[Image.fromarray(arr, "RGB")], return_tensors="pt")
        tensors[f"img{idx}.page"] = torch.from_numpy(arr.copy())
        tensors[f"img{idx}.pixel_values"] = out["pixel_values"].contiguous()
        meta.append({"seed": seed, "h": h, "w": w, "min_pixels": mn, "max_pixels": mx,
                     "grid_thw": out["image_grid_thw"][0].tolist()})
    save_file(tensors, os.path.join(GOLD, "image_kats.safetensors"))
    with open(os.path.join(GOLD, "image_kats.json"), "w") as f:
        json.dump({"source": "transformers Qwen2VLImageProcessorPil / smart_resize", "smart_resize": table, "images": meta,
                   "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]}, f,
                  indent=0)
    print("image_kats:", len(table), "smart_resize rows,", len(meta), "images")


# ------------------------------------------------------------------------------------------------ model
REP_PENALTY = 1.3  # strong enough to change several greedy choices of the tiny models

TINY = dict(
    vision=dict(depth=2, embed_dim=64, hidden_size=256, hidden_act="quick_gelu", mlp_ratio=2, num_heads=2, in_channels=3,
                patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
    text=dict(vocab_size=512, hidden_size=256, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
              num_key_value_heads=1, max_position_embeddings=4096, rms_norm_eps=1e-6, tie_word_embeddings=True,
              rope_parameters={"rope_type": "default", "rope_theta": 1000000.0, "mrope_section": [16, 24, 24]}),
    image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503, eos=510, pad=511)


# Qwen2.5-VL (the olmOCR-2 family, the reference's default OCR_MODEL): windows of 2x2 merged tokens, so the test pages
# have ragged windows on both axes; layer 1 of 3 attends over the whole page.  intermediate_size 88 is NOT a multiple
# of 64, like the real tower's 3420 (the engine zero-pads it).
TINY25 = dict(
    vision=dict(depth=3, hidden_size=64, out_hidden_size=256, hidden_act="silu", intermediate_size=88, num_heads=2,
                in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=56,
                fullatt_block_indexes=[1]),
    text=dict(TINY["text"]),
    image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503, eos=510, pad=511)


def build_tiny_hf(dtype: torch.dtype, family: str = "qwen2_vl"):
    if family == "qwen2_vl":
        from transformers import Qwen2VLConfig as Cfg, Qwen2VLForConditionalGeneration as Model
        spec = TINY
    else:
        from transformers import Qwen2_5_VLConfig as Cfg, Qwen2_5_VLForConditionalGeneration as Model
        spec = TINY25
    cfg = Cfg(vision_config=dict(spec["vision"]), text_config=dict(spec["text"]),
              image_token_id=spec["image_token_id"], video_token_id=spec["video_token_id"],
              vision_start_token_id=spec["vision_start_token_id"], vision_end_token_id=spec["vision_end_token_id"],
              tie_word_embeddings=True)
    torch.manual_seed(0)
    model = Model(cfg).eval()
    g = torch.Generator().manual_seed(20260504)
    with torch.no_grad():
        for name, prm in sorted(model.named_parameters()):
            if name.endswith("norm.weight") or "norm1.weight" in name or "norm2.weight" in name or "ln_q.weight" in name \
                    or "layernorm.weight" in name:
                v = 1.0 + 0.1 * torch.randn(prm.shape, generator=g)
            elif name.endswith(".bias"):
                v = 0.05 * torch.randn(prm.shape, generator=g)
            elif "embed_tokens" in name:
                v = 0.08 * torch.randn(prm.shape, generator=g)
            else:
                v = torch.randn(prm.shape, generator=g) * (1.5 / (prm.shape[-1] if prm.dim() < 3 else prm[0].numel()) ** 0.5)
            prm.copy_(v.to(torch.bfloat16).to(prm.dtype))  # bf16-representable, so one weight file serves both dtypes
    model.tie_weights()
    if dtype == torch.float32:
        return model, cfg
    # Load the way the reference does (from_pretrained(..., dtype=bf16), ocr_agent/tools.py:700-716): parameters become
    # bf16 but the rotary inv_freq buffers stay fp32.  `model.to(bf16)` would round those buffers too — an artefact no
    # real run has.
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        model.save_pretrained(tmp)
        model = Model.from_pretrained(tmp, dtype=dtype).eval()
    assert model.model.visual.rotary_pos_emb.inv_freq.dtype == torch.float32
    assert model.model.language_model.rotary_emb.inv_freq.dtype == torch.float32
    return model, cfg


def make_model(family: str = "qwen2_vl") -> None:
    from safetensors.torch import save_file
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from PIL import Image

    proc = Qwen2VLImageProcessorPil(min_pixels=28 * 28, max_pixels=1024 * 1024)
    cases = [("a", 11, (60, 90)), ("b", 12, (150, 200))]
    rng = np.random.default_rng(77)
    weights_saved = False
    spec, stem = (TINY, "qwen2vl_tiny") if family == "qwen2_vl" else (TINY25, "qwen25vl_tiny")
    meta = {"config": {k: v for k, v in spec.items()}, "cases": {}, "family": family,
            "source": "transformers %s, random init (seeded), greedy, min_new_tokens == max_new_tokens"
                      % ("Qwen2VLForConditionalGeneration" if family == "qwen2_vl" else "Qwen2_5_VLForConditionalGeneration")}
    N_NEW = 24
    for dtype, tag in [(torch.float32, "fp32"), (torch.bfloat16, "bf16")]:
        model, cfg = build_tiny_hf(dtype, family)
        model.generation_config.eos_token_id = TINY["eos"]
        model.generation_config.pad_token_id = TINY["pad"]
        if not weights_saved:
            sd = {k: v.to(torch.bfloat16).contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
            save_file(sd, os.path.join(GOLD, f"{stem}_weights.safetensors"))
            weights_saved = True
        tensors = {}
        for cname, seed, (h, w) in cases:
            page = make_page(seed, h, w)
            out = proc(images=[Image.fromarray(page, "RGB")], return_tensors="pt")
            pv, grid = out["pixel_values"], out["image_grid_thw"]
            n_img = int(grid[0].prod()) // 4
            crng = np.random.default_rng(1000 + seed)
            pre = crng.integers(0, 500, size=5).tolist()
            suf = crng.integers(0, 500, size=6).tolist()
            ids = pre + [TINY["vision_start_token_id"]] + [TINY["image_token_id"]] * n_img + [TINY["vision_end_token_id"]] + suf
            input_ids = torch.tensor([ids])
            mm = (input_ids == TINY["image_token_id"]).int()
            acts = {}

            def hook(name):
                def fn(mod, inp, outp):
                    acts[name] = (outp[0] if isinstance(outp, tuple) else outp).detach().clone()
                return fn

            vis = model.model.visual
            hs = [vis.patch_embed.register_forward_hook(hook("patch_embed")), vis.blocks[0].register_forward_hook(hook("vit_block0")),
                  vis.blocks[-1].register_forward_hook(hook("vit_last")),
                  # the tower's pooled output: for Qwen2.5-VL the merger rows put back in raster order
                  vis.register_forward_hook(lambda m_, i_, o_: acts.__setitem__("merger", o_.pooler_output.detach().clone())),
                  model.model.language_model.layers[0].register_forward_hook(hook("dec_layer0"))]
            with torch.no_grad():
                fw = model(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                           attention_mask=torch.ones_like(input_ids))
                pos, delta = model.model.get_rope_index(input_ids, mm, image_grid_thw=grid)
            for hdl in hs:
                hdl.remove()
            with torch.no_grad():
                gen = model.generate(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                                     attention_mask=torch.ones_like(input_ids), do_sample=False, max_new_tokens=N_NEW,
                                     min_new_tokens=N_NEW, output_logits=True, return_dict_in_generate=True)
            new = gen.sequences[0, input_ids.shape[1]:]
            # the deterministic part of the Qwen2.5-VL / olmOCR generation defaults: greedy with a repetition penalty
            with torch.no_grad():
                gen_rp = model.generate(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                                        attention_mask=torch.ones_like(input_ids), do_sample=False, max_new_tokens=N_NEW,
                                        min_new_tokens=N_NEW, repetition_penalty=REP_PENALTY, output_scores=True,
                                        return_dict_in_generate=True)
            tensors[f"{cname}.rp_tokens"] = gen_rp.sequences[0, input_ids.shape[1]:].to(torch.int32)
            tensors[f"{cname}.rp_scores"] = torch.stack([sc[0] for sc in gen_rp.scores]).float().contiguous()
            tensors[f"{cname}.page"] = torch.from_numpy(page.copy())
            tensors[f"{cname}.pixel_values"] = pv.contiguous()
            tensors[f"{cname}.input_ids"] = input_ids[0].to(torch.int32)
            tensors[f"{cname}.position_ids"] = pos[:, 0].to(torch.int32).contiguous()
            tensors[f"{cname}.prefill_logits"] = fw.logits[0].contiguous()
            tensors[f"{cname}.greedy_tokens"] = new.to(torch.int32)
            tensors[f"{cname}.step_logits"] = torch.stack([l[0] for l in gen.logits]).contiguous()
            for k, v in acts.items():
                tensors[f"{cname}.{k}"] = (v[0] if v.dim() == 3 else v).contiguous()
            meta["cases"][cname] = {"page_seed": seed, "page_hw": [h, w], "grid_thw": grid[0].tolist(), "n_new": N_NEW,
                                    "repetition_penalty": REP_PENALTY,
                                    "rope_delta": int(delta[0]), "T": len(ids)}
        save_file(tensors, os.path.join(GOLD, f"{stem}_{tag}.safetensors"))
        print(f"{stem}_{tag}.safetensors:", {k: tuple(v.shape) for k, v in tensors.items() if k.startswith("a.")})
    with open(os.path.join(GOLD, f"{stem}.json"), "w") as f:
        json.dump(meta, f, indent=0)
    for fn in os.listdir(GOLD):  # safetensors writes 0600; the GPU box reads the snapshot as another user
        os.chmod(os.path.join(GOLD, fn), 0o644)


# PaliGemma (BASELINE config 4): SigLIP head_dim 72 (8 heads: width 576 is a multiple of 64 like the real 1152), an MLP width
# that is not (600, real 4304) and Gemma head_dim 256 as in the 3B checkpoint, everything else tiny
TINYPG = dict(
    vision=dict(hidden_size=576, intermediate_size=600, num_hidden_layers=2, num_attention_heads=8, image_size=56,
                patch_size=14, projection_dim=256, vision_use_head=False, num_image_tokens=16),
    text=dict(model_type="gemma", vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
              num_attention_heads=2, num_key_value_heads=1, head_dim=256, max_position_embeddings=1024,
              hidden_act="gelu_pytorch_tanh"),
    image_token_id=500, eos=1, pad=0)


def make_model_paligemma() -> None:
    from safetensors.torch import save_file
    from transformers import PaliGemmaConfig, PaliGemmaForConditionalGeneration
    from transformers.models.siglip import SiglipImageProcessorPil
    from PIL import Image
    import tempfile

    spec = TINYPG
    S = spec["vision"]["image_size"]
    proc = SiglipImageProcessorPil(size={"height": S, "width": S}, resample=3, do_rescale=True, do_normalize=True,
                                   image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    meta = {"config": spec, "cases": {}, "family": "paligemma",
            "source": "transformers PaliGemmaForConditionalGeneration, random init (seeded), greedy, token_type_ids == 0 "
                      "(bidirectional image+prompt prefix), min_new_tokens == max_new_tokens"}
    N_NEW = 16
    weights_saved = False
    for dtype, tag in [(torch.float32, "fp32"), (torch.bfloat16, "bf16")]:
        cfg = PaliGemmaConfig(vision_config=dict(spec["vision"]), text_config=dict(spec["text"]),
                              image_token_id=spec["image_token_id"], projection_dim=256, hidden_size=256, vocab_size=512,
                              pad_token_id=spec["pad"], bos_token_id=2, eos_token_id=spec["eos"])
        torch.manual_seed(0)
        model = PaliGemmaForConditionalGeneration(cfg).eval()
        g = torch.Generator().manual_seed(20260505)
        with torch.no_grad():
            for name, prm in sorted(model.named_parameters()):
                if "layernorm.weight" in name and "language_model" in name or name.endswith("language_model.norm.weight"):
                    v = 0.1 * torch.randn(prm.shape, generator=g)          # Gemma norms multiply by (1 + w)
                elif "layer_norm" in name and name.endswith("weight") or "post_layernorm.weight" in name:
                    v = 1.0 + 0.1 * torch.randn(prm.shape, generator=g)
                elif name.endswith(".bias"):
                    v = 0.05 * torch.randn(prm.shape, generator=g)
                elif "embed_tokens" in name or "position_embedding" in name:
                    v = 0.08 * torch.randn(prm.shape, generator=g)
                else:
                    v = torch.randn(prm.shape, generator=g) * (1.2 / (prm.shape[-1] if prm.dim() < 3 else prm[0].numel()) ** 0.5)
                prm.copy_(v.to(torch.bfloat16).to(prm.dtype))
        model.tie_weights()
        if dtype != torch.float32:  # load like the reference does: bf16 parameters, fp32 rotary buffers
            with tempfile.TemporaryDirectory() as tmp:
                model.save_pretrained(tmp)
                model = PaliGemmaForConditionalGeneration.from_pretrained(tmp, dtype=dtype).eval()
            assert model.model.language_model.rotary_emb.inv_freq.dtype == torch.float32
        model.generation_config.eos_token_id = spec["eos"]
        model.generation_config.pad_token_id = spec["pad"]
        if not weights_saved:
            sd = {k: v.to(torch.bfloat16).contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
            save_file(sd, os.path.join(GOLD, "paligemma_tiny_weights.safetensors"))
            weights_saved = True
        tensors = {}
        for cname, seed, (h, w) in [("a", 21, (60, 90)), ("b", 22, (150, 200))]:
            page = make_page(seed, h, w)
            pv = proc(images=[Image.fromarray(page, "RGB")], return_tensors="pt")["pixel_values"]
            n_img = (S // spec["vision"]["patch_size"]) ** 2
            crng = np.random.default_rng(2000 + seed)
            ids = [spec["image_token_id"]] * n_img + [2] + crng.integers(3, 490, size=7).tolist()
            input_ids = torch.tensor([ids])
            tt = torch.zeros_like(input_ids)
            acts = {}

            def hook(name):
                def fn(mod, inp, outp):
                    acts[name] = (outp[0] if isinstance(outp, tuple) else outp).detach().clone()
                return fn

            vis = model.model.vision_tower
            hs = [vis.embeddings.register_forward_hook(hook("patch_embed")),
                  vis.encoder.layers[0].register_forward_hook(hook("vit_block0")),
                  vis.post_layernorm.register_forward_hook(hook("vit_last")),
                  model.model.multi_modal_projector.register_forward_hook(hook("projector")),
                  model.model.language_model.layers[0].register_forward_hook(hook("dec_layer0"))]
            with torch.no_grad():
                fw = model(input_ids=input_ids, pixel_values=pv.to(dtype), token_type_ids=tt,
                           attention_mask=torch.ones_like(input_ids))
            for hdl in hs:
                hdl.remove()
            with torch.no_grad():
                gen = model.generate(input_ids=input_ids, pixel_values=pv.to(dtype), token_type_ids=tt,
                                     attention_mask=torch.ones_like(input_ids), do_sample=False, max_new_tokens=N_NEW,
                                     min_new_tokens=N_NEW, output_logits=True, return_dict_in_generate=True)
            tensors[f"{cname}.page"] = torch.from_numpy(page.copy())
            tensors[f"{cname}.pixel_values"] = pv[0].contiguous()
            tensors[f"{cname}.input_ids"] = input_ids[0].to(torch.int32)
            tensors[f"{cname}.prefill_logits"] = fw.logits[0].contiguous()
            tensors[f"{cname}.greedy_tokens"] = gen.sequences[0, input_ids.shape[1]:].to(torch.int32)
            tensors[f"{cname}.step_logits"] = torch.stack([l[0] for l in gen.logits]).contiguous()
            for k, v in acts.items():
                tensors[f"{cname}.{k}"] = (v[0] if v.dim() == 3 else v).contiguous()
            meta["cases"][cname] = {"page_seed": seed, "page_hw": [h, w], "n_new": N_NEW, "T": len(ids)}
        save_file(tensors, os.path.join(GOLD, f"paligemma_tiny_{tag}.safetensors"))
        print(f"paligemma_tiny_{tag}.safetensors:", {k: tuple(v.shape) for k, v in tensors.items() if k.startswith("a.")})
    with open(os.path.join(GOLD, "paligemma_tiny.json"), "w") as f:
        json.dump(meta, f, indent=0)
    for fn in os.listdir(GOLD):
        os.chmod(os.path.join(GOLD, fn), 0o644)


# ------------------------------------------------------------------------------------------------ processor / tokenizer
# the chat template the Qwen2-VL / Qwen2.5-VL / olmOCR-2 checkpoints ship (chat_template.json of the public model cards)
QWEN2VL_CHAT_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% set video_count = namespace(value=0) %}{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n{% else %}"
    "{% for content in message['content'] %}{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}"
    "<|vision_start|><|image_pad|><|vision_end|>{% elif content['type'] == 'video' or 'video' in content %}"
    "{% set video_count.value = video_count.value + 1 %}{% if add_vision_id %}Video {{ video_count.value }}: {% endif %}"
    "<|vision_start|><|video_pad|><|vision_end|>{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n"
    "{% endif %}{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")
TOK_SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|vision_pad|>",
                "<|image_pad|>", "<|video_pad|>"]


def make_tokenizer() -> None:
    """tests/golden/tokenizer_tiny/: a checkpoint-shaped directory (tokenizer.json, tokenizer_config.json, chat template,
    preprocessor_config.json) saved by HF's own save_pretrained; tests/golden/tokenizer_kats.json: what
    processor.apply_chat_template / processor.decode return for it — the calls of ocr_agent/tools.py:756-769."""
    from PIL import Image
    from tokenizers import AddedToken, Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    # byte-level BPE built like Qwen2's tokenizer.json: NFC, the GPT-4-style split pattern, ByteLevel, no prefix space
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    pat = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    tok.post_processor = processors.ByteLevel(trim_offsets=False)
    rng = random.Random(11)
    corpus = ["Extract and return all the text from this handwritten document.", "You are a helpful assistant.",
              "system user assistant", "Dear Anna, thank you for the letter — it arrived on 3 May. “Quoted” text, naïve café."]
    corpus += [_rand_text(rng, rng.randint(5, 30)) for _ in range(60)]
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=480, special_tokens=[], show_progress=False,
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in TOK_SPECIALS])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>",
                                   additional_special_tokens=TOK_SPECIALS[1:])

    # Qwen2VLProcessor itself cannot be constructed here (it insists on a video processor, whose class needs torchvision —
    # absent), so its three steps are driven one by one on HF's own objects: the chat-template renderer
    # (tokenizer.apply_chat_template -> utils/chat_template_utils.render_jinja_template, the function the processor calls), the
    # placeholder expansion `image_token * (grid.prod() // merge^2)` (processing_qwen2_vl.py:58-61, the one restated line) with
    # the grid from HF's PIL image processor, and tokenizer.__call__.
    fast.chat_template = QWEN2VL_CHAT_TEMPLATE
    ip = Qwen2VLImageProcessorPil(min_pixels=28 * 28, max_pixels=1024 * 1024)
    out_dir = os.path.join(GOLD, "tokenizer_tiny")
    os.makedirs(out_dir, exist_ok=True)
    for fn in os.listdir(out_dir):
        os.remove(os.path.join(out_dir, fn))
    fast.save_pretrained(out_dir)
    ip.save_pretrained(out_dir)
    ids = {t: tok.token_to_id(t) for t in TOK_SPECIALS}
    cases = []
    prompts = ["Extract and return all the text from this handwritten document.", "Read the page.\nKeep line breaks!",
               "Qu’est-ce que c’est — naïve café? 日本語 42"]
    for i, ((h, w), prompt) in enumerate(zip(((84, 112), (56, 56), (300, 140)), prompts)):
        img = Image.fromarray(make_page(i, h, w), "RGB")
        messages = [{"role": "user", "content": [{"type": "image", "url": f"p{i}.png"}, {"type": "text", "text": prompt}]}]
        text = fast.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
        grid = ip(images=[img], return_tensors="pt")["image_grid_thw"][0]
        n_tok = int(grid.prod()) // ip.merge_size ** 2
        assert text.count("<|image_pad|>") == 1
        expanded = text.replace("<|image_pad|>", "<|image_pad|>" * n_tok)
        cases.append({"page_hw": [h, w], "prompt": prompt, "rendered": text, "image_grid_thw": grid.tolist(),
                      "input_ids": fast(expanded)["input_ids"]})
    proc = fast  # processor.decode forwards to tokenizer.decode (processing_utils.py:1939-1946)
    dec = []
    texts = ["Dear Anna, thank you for the letter.", "line one\nline two\n\n  indented", "café — “quoted” 日本",
             " leading space and trailing space ", "it 's a test , really !", ""]
    for t in texts:
        body = tok.encode(t, add_special_tokens=False).ids
        for tail in ([], [ids["<|im_end|>"]], [ids["<|im_end|>"], ids["<|endoftext|>"], ids["<|endoftext|>"]]):
            seq = body + tail
            dec.append({"ids": seq, "skip": proc.decode(seq, skip_special_tokens=True), "keep": proc.decode(seq, skip_special_tokens=False)})
    # truncated multi-byte characters (a generation cut by max_new_tokens) and specials inside the stream
    cut = tok.encode("日本語", add_special_tokens=False).ids[:-1]
    for seq in (cut, [ids["<|vision_start|>"]] + tok.encode("mid", add_special_tokens=False).ids + [ids["<|vision_end|>"]] + cut):
        dec.append({"ids": seq, "skip": proc.decode(seq, skip_special_tokens=True), "keep": proc.decode(seq, skip_special_tokens=False)})
    enc_kats = [{"text": t, "ids": fast(t)["input_ids"]} for t in texts + prompts + ["<|im_start|>user\n<|vision_start|><|image_pad|><|image_pad|><|vision_end|>x<|im_end|>\n"]]
    # PaliGemma (BASELINE config 4): its processor has no chat template; it builds "<image> x n <bos> prompt \n"
    # (processing_paligemma.py build_string_from_input).  HF's own PaliGemmaProcessor is constructible offline: run it.
    from transformers.models.paligemma.processing_paligemma import PaliGemmaProcessor
    from transformers.models.siglip import SiglipImageProcessorPil

    ptok = Tokenizer(models.BPE())
    ptok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    ptok.decoder = decoders.ByteLevel()
    ptok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=420, special_tokens=["<pad>", "<eos>", "<bos>"], show_progress=False,
                                                         initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    pfast = PreTrainedTokenizerFast(tokenizer_object=ptok, bos_token="<bos>", eos_token="<eos>", pad_token="<pad>")
    S = 56
    pip_ = SiglipImageProcessorPil(size={"height": S, "width": S}, resample=3, do_rescale=True, do_normalize=True,
                                   image_mean=[0.5] * 3, image_std=[0.5] * 3)
    pip_.image_seq_length = (S // 14) ** 2
    pproc = PaliGemmaProcessor(image_processor=pip_, tokenizer=pfast)   # adds "<image>" to the tokenizer
    pg_dir = os.path.join(GOLD, "tokenizer_pg_tiny")
    os.makedirs(pg_dir, exist_ok=True)
    for fn in os.listdir(pg_dir):
        os.remove(os.path.join(pg_dir, fn))
    pproc.tokenizer.save_pretrained(pg_dir)
    pg_cases = []
    for i, prompt in enumerate(prompts):
        img = Image.fromarray(make_page(10 + i, 80, 100), "RGB")
        enc = pproc(text="<image>" + prompt, images=img, return_tensors="pt")
        pg_cases.append({"prompt": prompt, "input_ids": enc["input_ids"][0].tolist(), "image_tokens": pip_.image_seq_length})
    pg = {"image_token_id": pproc.image_token_id, "bos_token_id": pfast.bos_token_id, "eos_token_id": pfast.eos_token_id,
          "pad_token_id": pfast.pad_token_id, "chat": pg_cases,
          "decode": [{"ids": c["input_ids"][c["image_tokens"]:] + [pfast.eos_token_id],
                      "skip": pproc.decode(c["input_ids"][c["image_tokens"]:] + [pfast.eos_token_id], skip_special_tokens=True)} for c in pg_cases]}
    with open(os.path.join(GOLD, "tokenizer_kats.json"), "w", encoding="utf-8") as f:
        json.dump({"source": "transformers PreTrainedTokenizerFast(tiny byte-level BPE).apply_chat_template(Qwen2-VL template) / __call__ / "
                             "decode + Qwen2VLImageProcessorPil grids; the <|image_pad|> expansion is Qwen2VLProcessor.replace_image_token restated",
                   "special_ids": ids, "chat": cases, "decode": dec, "encode": enc_kats,
                   "paligemma": dict(pg, source="transformers PaliGemmaProcessor(SiglipImageProcessorPil, PreTrainedTokenizerFast(tiny "
                                                "byte-level BPE)).__call__(text='<image>' + prompt, images=page) / .decode")},
                  f, indent=0, ensure_ascii=True)


# ------------------------------------------------------------------------------------------------ trained tiny checkpoints
# The accuracy bar of the path ("CER within 0.5 % of the reference", BASELINE.json; metric = cer() over the text generate()
# returns, ocr_agent/tools.py:103-139, :764-769) needs a model whose greedy choices are DECISIVE: on the random-init goldens
# above the logits are nearly tied and a free-running stream diverges on the first 1-ulp difference, which says nothing.
# So: the same tiny architectures, briefly trained here on CPU to transcribe 12 synthetic pages (page -> its own sentence;
# all sentences share their first words, so the branch is decided by the image), saved as a complete checkpoint DIRECTORY
# (config.json, model.safetensors, generation_config.json, the tokenizer_tiny files) and then read exactly as the reference
# reads its model: from_pretrained(dtype=bfloat16) (tools.py:700-709), chat template on run_ocr's message list, the image
# processor with the reference's pixel bounds (config.py:17-18), generate(**inputs, max_new_tokens=...) (tools.py:764-765),
# decode(new tokens, skip_special_tokens=True) (tools.py:767-769).
TRAINED_PAGES = [(100, (300, 260)), (101, (260, 300)), (102, (280, 280)), (103, (320, 250)), (104, (300, 260)), (105, (256, 310)),
                 (106, (280, 280)), (107, (300, 260)), (108, (260, 300)), (109, (320, 250)), (110, (280, 280)), (111, (256, 310))]
# paper colours (synth.tint_page): make_page's scribbles differ patch by patch only, and a decoder that attends evenly over the
# image tokens sees the same average for every page; real scans differ globally too
TRAINED_TINTS = [(256, 256, 256), (256, 232, 176), (184, 224, 256), (232, 256, 184), (256, 192, 208), (200, 200, 200),
                 (256, 256, 160), (176, 256, 232), (224, 184, 256), (256, 216, 216), (168, 208, 168), (216, 216, 256)]
TRAINED_EVAL = 8          # the first 8 pages carry HF streams
TRAINED_MAX_NEW = 128
TRAINED_LR = float(os.environ.get("HWOCR_TRAINED_LR", "2e-3"))
TRAINED_STEPS = int(os.environ.get("HWOCR_TRAINED_STEPS", "110"))
TRAINED_HEAD_WEIGHT = float(os.environ.get("HWOCR_TRAINED_HEADW", "10"))  # loss weight of a sentence's first tokens: only there does the answer depend on the IMAGE alone


def _trained_texts(tok_len) -> list[str]:
    """One sentence per page, each starting with a word of its own (the page alone decides the first token); every other one is
    long enough to be cut by the 128-token budget, the rest stop on EOS."""
    rng = random.Random(4242)
    heads = rng.sample(sorted(set(WORDS)), len(TRAINED_PAGES))
    out = []
    for i in range(len(TRAINED_PAGES)):
        want = 150 if i % 2 else rng.randint(40, 100)
        words = [heads[i].capitalize()]
        while True:
            words.append(rng.choice(WORDS) + (rng.choice(",.;") if rng.random() < 0.12 else ""))
            t = " ".join(words)
            if tok_len(t) >= want:
                break
        out.append(t)
    return out


def make_trained(family: str = "qwen2_vl") -> None:
    import shutil
    import tempfile

    from PIL import Image
    from safetensors.torch import save_file
    from transformers import GenerationConfig, PreTrainedTokenizerFast
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    if family == "qwen2_vl":
        from transformers import Qwen2VLConfig as Cfg, Qwen2VLForConditionalGeneration as Model
        spec, stem = TINY, "trained_qwen2vl"
    else:
        from transformers import Qwen2_5_VLConfig as Cfg, Qwen2_5_VLForConditionalGeneration as Model
        spec, stem = TINY25, "trained_qwen25vl"
    tok_dir = os.path.join(GOLD, "tokenizer_tiny")
    fast = PreTrainedTokenizerFast.from_pretrained(tok_dir)
    sid = {t: fast.convert_tokens_to_ids(t) for t in TOK_SPECIALS}
    eos, pad, img_id = sid["<|im_end|>"], sid["<|endoftext|>"], sid["<|image_pad|>"]
    # the reference's processor bounds (ocr_agent/tools.py:700-704, config.py:17-18)
    ip = Qwen2VLImageProcessorPil(min_pixels=256 * 256, max_pixels=1024 * 1024)
    cfg = Cfg(vision_config=dict(spec["vision"]), text_config=dict(spec["text"]), image_token_id=img_id,
              video_token_id=sid["<|video_pad|>"], vision_start_token_id=sid["<|vision_start|>"],
              vision_end_token_id=sid["<|vision_end|>"], tie_word_embeddings=True)
    threads = torch.get_num_threads()
    torch.set_num_threads(int(os.environ.get("HWOCR_TRAINED_THREADS", "1")))  # 1: one summation order whatever the host: the fixture regenerates byte for byte
    torch.use_deterministic_algorithms(True)
    torch.manual_seed(0)
    model = Model(cfg)
    prompt = "Extract and return all the text from this handwritten document."  # config.OCR_PROMPT (config.py:20)
    texts = _trained_texts(lambda t: len(fast(t)["input_ids"]))
    items = []
    for (seed, (h, w)), tint, text in zip(TRAINED_PAGES, TRAINED_TINTS, texts):
        page = tint_page(make_page(seed, h, w), tint)
        out = ip(images=[Image.fromarray(page, "RGB")], return_tensors="pt")
        grid = out["image_grid_thw"]
        messages = [{"role": "user", "content": [{"type": "image", "url": f"page{seed}.png"}, {"type": "text", "text": prompt}]}]
        rendered = fast.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
        n_tok = int(grid[0].prod()) // ip.merge_size ** 2
        ids = fast(rendered.replace("<|image_pad|>", "<|image_pad|>" * n_tok))["input_ids"]   # processing_qwen2_vl.py:58-61
        ans = fast(text)["input_ids"] + [eos]
        items.append(dict(seed=seed, hw=(h, w), tint=tint, pv=out["pixel_values"], grid=grid, ids=ids, ans=ans, text=text))
    opt = torch.optim.AdamW(model.parameters(), lr=TRAINED_LR, weight_decay=0.0)
    model.train()
    for step in range(TRAINED_STEPS):
        for g_ in opt.param_groups:
            g_["lr"] = TRAINED_LR * min(1.0, (step + 1) / 20) * (0.5 * (1 + np.cos(np.pi * step / TRAINED_STEPS)) * 0.95 + 0.05)
        opt.zero_grad()
        total = 0.0
        for it in items:
            full = torch.tensor([it["ids"] + it["ans"]])
            mm = (full == img_id).int()
            out = model(input_ids=full, pixel_values=it["pv"], image_grid_thw=it["grid"], mm_token_type_ids=mm,
                        attention_mask=torch.ones_like(full))
            n0 = len(it["ids"])
            ce = torch.nn.functional.cross_entropy(out.logits[0, n0 - 1:-1].float(), full[0, n0:], reduction="none")
            wgt = torch.ones_like(ce)
            wgt[:4] = TRAINED_HEAD_WEIGHT
            loss = (ce * wgt).sum() / wgt.sum()
            (loss / len(items)).backward()
            total += float(ce.mean().detach())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if step % 20 == 0 or step == TRAINED_STEPS - 1:
            print(f"  [{stem}] step {step}: loss {total / len(items):.4f}", flush=True)
    model.eval()
    out_dir = os.path.join(GOLD, stem)
    if os.path.isdir(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    with torch.no_grad():
        for prm in model.parameters():
            prm.copy_(prm.to(torch.bfloat16).to(prm.dtype))
    model.tie_weights()
    cfg.save_pretrained(out_dir)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
    save_file(sd, os.path.join(out_dir, "model.safetensors"), metadata={"format": "pt"})
    # what `generate(**inputs, max_new_tokens=...)` runs with (tools.py:765): the checkpoint's generation defaults
    GenerationConfig(do_sample=False, eos_token_id=[eos, pad], pad_token_id=pad, bos_token_id=pad).save_pretrained(out_dir)
    for fn in ("tokenizer.json", "tokenizer_config.json", "chat_template.jinja"):
        shutil.copyfile(os.path.join(tok_dir, fn), os.path.join(out_dir, fn))
    ip.save_pretrained(out_dir)
    # read it back the way the reference does, and transcribe
    hf = Model.from_pretrained(out_dir, dtype=torch.bfloat16).eval()
    assert hf.model.language_model.rotary_emb.inv_freq.dtype == torch.float32
    assert hf.generation_config.eos_token_id == [eos, pad] and hf.generation_config.do_sample is False
    cases = []
    margins_all = []
    for it in items[:TRAINED_EVAL]:
        input_ids = torch.tensor([it["ids"]])
        mm = (input_ids == img_id).int()
        with torch.no_grad():
            gen = hf.generate(input_ids=input_ids, pixel_values=it["pv"], image_grid_thw=it["grid"], mm_token_type_ids=mm,
                              attention_mask=torch.ones_like(input_ids), max_new_tokens=TRAINED_MAX_NEW,
                              output_logits=True, return_dict_in_generate=True)
        new = gen.sequences[0, input_ids.shape[1]:].tolist()
        logits = torch.stack([l[0] for l in gen.logits]).float()
        top2 = logits.topk(2, dim=-1).values
        margins = (top2[:, 0] - top2[:, 1]).tolist()
        margins_all += margins
        hf_text = fast.decode(new, skip_special_tokens=True)
        cases.append({"page_seed": it["seed"], "page_hw": list(it["hw"]), "page_tint": list(it["tint"]), "grid_thw": it["grid"][0].tolist(),
                      "input_ids": it["ids"], "trained_on": it["text"], "hf_tokens": new, "hf_text": hf_text,
                      "margins": [round(m, 4) for m in margins], "stopped_on_eos": new[-1] in (eos, pad)})
        print(f"  [{stem}] page {it['seed']}: {len(new)} tokens, min margin {min(margins):.2f}, "
              f"cer vs trained text {_import_reference_tools().cer(it['text'], hf_text):.4f}")
    decisive = float(np.mean([m > 1.0 for m in margins_all]))
    assert decisive >= 0.95, f"greedy decoding is not decisive: margin > 1.0 on {decisive:.3f} of the steps"
    with open(os.path.join(GOLD, stem + ".json"), "w", encoding="utf-8") as f:
        json.dump({"source": f"transformers {Model.__name__}: {TRAINED_STEPS} AdamW steps on CPU (1 thread, seeded) on synth.make_page "
                             "pages -> sentences, saved with save_pretrained, read back with from_pretrained(dtype=bfloat16) and "
                             "generate(**inputs, max_new_tokens=128) under the saved generation_config (greedy), decoded with the "
                             "checkpoint's tokenizer (skip_special_tokens=True): the call sequence of ocr_agent/tools.py:700-709, :744-769",
                   "family": family, "checkpoint_dir": stem, "prompt": prompt, "max_new_tokens": TRAINED_MAX_NEW,
                   "min_pixels": 256 * 256, "max_pixels": 1024 * 1024, "eos_token_id": [eos, pad], "pad_token_id": pad,
                   "decisive_fraction_margin_gt_1": round(decisive, 4), "cases": cases}, f, indent=0, ensure_ascii=True)
    for root, _, files in os.walk(out_dir):
        for fn in files:
            os.chmod(os.path.join(root, fn), 0o644)
    os.chmod(os.path.join(GOLD, stem + ".json"), 0o644)
    torch.use_deterministic_algorithms(False)
    torch.set_num_threads(threads)
    print(f"{stem}: decisive on {decisive:.3f} of {len(margins_all)} steps; {sum(c['stopped_on_eos'] for c in cases)} of "
          f"{len(cases)} streams stop on EOS")


def make_trained_paligemma() -> None:
    """The same for BASELINE config 4's family in bf16: a briefly trained tiny PaliGemma (SigLIP head_dim 72, Gemma head_dim 256)
    as a checkpoint directory, read by HF's own PaliGemmaProcessor ("<image>" x n + <bos> + prompt + newline, bidirectional prefix)
    and generate().  Pages are squashed to 56 x 56 by the SigLIP image processor: the paper tint is what survives."""
    import shutil

    from PIL import Image
    from safetensors.torch import save_file
    from transformers import GenerationConfig, PaliGemmaConfig, PaliGemmaForConditionalGeneration, PreTrainedTokenizerFast
    from transformers.models.paligemma.processing_paligemma import PaliGemmaProcessor
    from transformers.models.siglip import SiglipImageProcessorPil

    spec, stem = TINYPG, "trained_paligemma"
    S = spec["vision"]["image_size"]
    fast = PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLD, "tokenizer_pg_tiny"))
    pip_ = SiglipImageProcessorPil(size={"height": S, "width": S}, resample=3, do_rescale=True, do_normalize=True,
                                   image_mean=[0.5] * 3, image_std=[0.5] * 3)
    pip_.image_seq_length = (S // spec["vision"]["patch_size"]) ** 2
    pproc = PaliGemmaProcessor(image_processor=pip_, tokenizer=fast)
    img_id, bos, eos, pad = pproc.image_token_id, fast.bos_token_id, fast.eos_token_id, fast.pad_token_id
    cfg = PaliGemmaConfig(vision_config=dict(spec["vision"]), text_config=dict(spec["text"]), image_token_id=img_id, projection_dim=256,
                          hidden_size=256, vocab_size=512, pad_token_id=pad, bos_token_id=bos, eos_token_id=eos)
    threads = torch.get_num_threads()
    torch.set_num_threads(int(os.environ.get("HWOCR_TRAINED_THREADS", "1")))
    torch.use_deterministic_algorithms(True)
    torch.manual_seed(0)
    model = PaliGemmaForConditionalGeneration(cfg)
    prompt = "Extract and return all the text from this handwritten document."
    texts = _trained_texts(lambda t: len(fast(t, add_special_tokens=False)["input_ids"]))
    items = []
    for (seed, (h, w)), tint, text in zip(TRAINED_PAGES, TRAINED_TINTS, texts):
        img = Image.fromarray(tint_page(make_page(seed, h, w), tint), "RGB")
        enc = pproc(text="<image>" + prompt, images=img, return_tensors="pt")            # what run_ocr's processor call yields
        ans = fast(text, add_special_tokens=False)["input_ids"] + [eos]
        items.append(dict(seed=seed, hw=(h, w), tint=tint, pv=enc["pixel_values"], ids=enc["input_ids"][0].tolist(), ans=ans, text=text))
    steps = int(os.environ.get("HWOCR_TRAINED_STEPS_PG", "160"))
    opt = torch.optim.AdamW(model.parameters(), lr=TRAINED_LR, weight_decay=0.0)
    model.train()
    for step in range(steps):
        for g_ in opt.param_groups:
            g_["lr"] = TRAINED_LR * min(1.0, (step + 1) / 20) * (0.5 * (1 + np.cos(np.pi * step / steps)) * 0.95 + 0.05)
        opt.zero_grad()
        total = 0.0
        for it in items:
            full = torch.tensor([it["ids"] + it["ans"]])
            n0 = len(it["ids"])
            tt = torch.zeros_like(full)
            tt[0, n0:] = 1                                                   # the prefix is bidirectional, the answer causal
            out = model(input_ids=full, pixel_values=it["pv"], token_type_ids=tt, attention_mask=torch.ones_like(full),
                        labels=torch.where(tt == 1, full, torch.full_like(full, -100)))
            ce = torch.nn.functional.cross_entropy(out.logits[0, n0 - 1:-1].float(), full[0, n0:], reduction="none")
            wgt = torch.ones_like(ce)
            wgt[:4] = TRAINED_HEAD_WEIGHT
            ((ce * wgt).sum() / wgt.sum() / len(items)).backward()
            total += float(ce.mean().detach())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if step % 20 == 0 or step == steps - 1:
            print(f"  [{stem}] step {step}: loss {total / len(items):.4f}", flush=True)
    model.eval()
    out_dir = os.path.join(GOLD, stem)
    if os.path.isdir(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    with torch.no_grad():
        for prm in model.parameters():
            prm.copy_(prm.to(torch.bfloat16).to(prm.dtype))
    model.tie_weights()
    cfg.save_pretrained(out_dir)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
    save_file(sd, os.path.join(out_dir, "model.safetensors"), metadata={"format": "pt"})
    GenerationConfig(do_sample=False, eos_token_id=eos, pad_token_id=pad, bos_token_id=bos).save_pretrained(out_dir)
    pproc.tokenizer.save_pretrained(out_dir)
    pip_.save_pretrained(out_dir)
    hf = PaliGemmaForConditionalGeneration.from_pretrained(out_dir, dtype=torch.bfloat16).eval()
    assert hf.model.language_model.rotary_emb.inv_freq.dtype == torch.float32
    cases, margins_all = [], []
    for it in items[:TRAINED_EVAL]:
        input_ids = torch.tensor([it["ids"]])
        with torch.no_grad():
            gen = hf.generate(input_ids=input_ids, pixel_values=it["pv"].to(torch.bfloat16), token_type_ids=torch.zeros_like(input_ids),
                              attention_mask=torch.ones_like(input_ids), max_new_tokens=TRAINED_MAX_NEW, output_logits=True,
                              return_dict_in_generate=True)
        new = gen.sequences[0, input_ids.shape[1]:].tolist()
        logits = torch.stack([l[0] for l in gen.logits]).float()
        top2 = logits.topk(2, dim=-1).values
        margins = (top2[:, 0] - top2[:, 1]).tolist()
        margins_all += margins
        hf_text = pproc.decode(new, skip_special_tokens=True)
        cases.append({"page_seed": it["seed"], "page_hw": list(it["hw"]), "page_tint": list(it["tint"]), "input_ids": it["ids"],
                      "trained_on": it["text"], "hf_tokens": new, "hf_text": hf_text, "margins": [round(m, 4) for m in margins],
                      "stopped_on_eos": new[-1] in (eos, pad)})
        print(f"  [{stem}] page {it['seed']}: {len(new)} tokens, min margin {min(margins):.2f}, "
              f"cer vs trained text {_import_reference_tools().cer(it['text'], hf_text):.4f}")
    decisive = float(np.mean([m > 1.0 for m in margins_all]))
    assert decisive >= 0.95, f"greedy decoding is not decisive: margin > 1.0 on {decisive:.3f} of the steps"
    with open(os.path.join(GOLD, stem + ".json"), "w", encoding="utf-8") as f:
        json.dump({"source": f"transformers PaliGemmaForConditionalGeneration: {steps} AdamW steps on CPU (1 thread, seeded) on tinted synth.make_page "
                             "pages -> sentences, saved with save_pretrained, read back with from_pretrained(dtype=bfloat16), inputs from HF's "
                             "PaliGemmaProcessor(text='<image>' + prompt, images=page), generate(max_new_tokens=128) under the saved "
                             "generation_config (greedy), decoded with the checkpoint's tokenizer (skip_special_tokens=True)",
                   "family": "paligemma", "checkpoint_dir": stem, "prompt": prompt, "max_new_tokens": TRAINED_MAX_NEW, "image_size": S,
                   "eos_token_id": [eos], "pad_token_id": pad, "decisive_fraction_margin_gt_1": round(decisive, 4), "cases": cases},
                  f, indent=0, ensure_ascii=True)
    for root, _, files in os.walk(out_dir):
        for fn in files:
            os.chmod(os.path.join(root, fn), 0o644)
    os.chmod(os.path.join(GOLD, stem + ".json"), 0o644)
    torch.use_deterministic_algorithms(False)
    torch.set_num_threads(threads)
    print(f"{stem}: decisive on {decisive:.3f} of {len(margins_all)} steps; {sum(c['stopped_on_eos'] for c in cases)} of {len(cases)} streams stop on EOS")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="text,preprocess,nodes,image,model,model25,paligemma,tokenizer,trained,trained25,trainedpg")
    only = set(ap.parse_args().only.split(","))
    os.makedirs(GOLD, exist_ok=True)
    if only & {"text", "preprocess"}:
        tools = _import_reference_tools()
        if "text" in only:
            make_text(tools)
        if "preprocess" in only:
            make_preprocess(tools)
    if "nodes" in only:
        make_nodes()
    if "image" in only:
        make_image()
    if "model" in only:
        make_model("qwen2_vl")
    if "model25" in only:
        make_model("qwen2_5_vl")
    if "paligemma" in only:
        make_model_paligemma()
    if "tokenizer" in only:
        make_tokenizer()
    if "trained" in only:      # after the tokenizer leg: the checkpoint directories carry tokenizer_tiny's files
        make_trained("qwen2_vl")
    if "trained25" in only:
        make_trained("qwen2_5_vl")
    if "trainedpg" in only:
        make_trained_paligemma()


if __name__ == "__main__":
    main()
