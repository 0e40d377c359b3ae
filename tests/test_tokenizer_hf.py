"""Checkpoint tokenizer + chat template against outputs of HF's own objects.

tests/golden/tokenizer_tiny/ is a checkpoint-shaped directory written by HF's save_pretrained (tokenizer.json,
tokenizer_config.json, chat_template.jinja, preprocessor_config.json) around a tiny byte-level BPE built like Qwen2's;
tests/golden/tokenizer_kats.json holds what HF's PreTrainedTokenizerFast.apply_chat_template / __call__ / decode return
for it (tools/make_goldens.py make_tokenizer).  Nothing here imports transformers."""
import json
import os
import shutil

import numpy as np
import pytest

from handwritten_ocr_amd import engine, imageproc, tokenizer
from tests._golden import GOLD, load_json

TOKDIR = os.path.join(GOLD, "tokenizer_tiny")


@pytest.fixture(scope="module")
def kats():
    return load_json("tokenizer_kats.json")


@pytest.fixture(scope="module")
def proc(kats):
    sp = kats["special_ids"]
    cfg = engine.preset("tiny")
    cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id = sp["<|image_pad|>"], sp["<|vision_start|>"], sp["<|vision_end|>"]
    cfg.im_start_id, cfg.im_end_id, cfg.eos_ids, cfg.pad_id = sp["<|im_start|>"], sp["<|im_end|>"], (sp["<|im_end|>"],), sp["<|endoftext|>"]
    return tokenizer.Processor(cfg, tokenizer.HFTokenizer(cfg, TOKDIR), template_dir=TOKDIR)


def test_template_is_found_and_rendered_like_hf(proc, kats):
    assert proc.template is not None and "<|im_start|>" in proc.template.source
    for c in kats["chat"]:
        assert proc.chat_text(c["prompt"]) == c["rendered"]


def test_chat_ids_equal_hf_prompt_ids(proc, kats):
    cfg = proc.cfg
    for c in kats["chat"]:
        h, w = c["page_hw"]
        th, tw = imageproc.smart_resize(h, w, cfg.patch_size * cfg.merge, cfg.min_pixels, cfg.max_pixels)
        assert [1, th // cfg.patch_size, tw // cfg.patch_size] == c["image_grid_thw"]
        n = (th // cfg.patch_size) * (tw // cfg.patch_size) // cfg.merge ** 2
        ids = proc.chat_ids(c["prompt"], n)
        assert ids.dtype == np.int32 and ids.tolist() == c["input_ids"]


def test_encode_equals_hf(proc, kats):
    for k in kats["encode"]:
        assert proc.tokenizer.encode(k["text"]) == k["ids"], k["text"]


def test_decode_equals_hf(proc, kats):
    """processor.decode(new_ids, skip_special_tokens=True) of tools.py:767-769 — EOS / pad tails, specials inside the
    stream, whitespace runs, a multi-byte character cut by the token budget."""
    for k in kats["decode"]:
        assert proc.decode(k["ids"], skip_special_tokens=True) == k["skip"]
        assert proc.decode(k["ids"], skip_special_tokens=False) == k["keep"]


@pytest.mark.parametrize("where", ["chat_template.json", "processor_config.json", "tokenizer_config.json"])
def test_template_locations(tmp_path, where, kats, proc):
    """Older checkpoints keep the template in chat_template.json (the Qwen2-VL model cards) or inside a config file."""
    d = tmp_path / "ckpt"
    shutil.copytree(TOKDIR, d)
    src = (d / "chat_template.jinja").read_text(encoding="utf-8")
    (d / "chat_template.jinja").unlink()
    target = d / where
    body = json.loads(target.read_text()) if target.exists() else {}
    body["chat_template"] = src
    target.write_text(json.dumps(body))
    t = tokenizer.ChatTemplate.from_dir(str(d))
    assert t is not None and t.source == src
    p2 = tokenizer.Processor(proc.cfg, proc.tokenizer, template_dir=str(d))
    assert p2.chat_text(kats["chat"][0]["prompt"]) == kats["chat"][0]["rendered"]


def test_no_template_keeps_builtin_layout(tmp_path, proc):
    d = tmp_path / "ckpt"
    shutil.copytree(TOKDIR, d)
    (d / "chat_template.jinja").unlink()
    assert tokenizer.ChatTemplate.from_dir(str(d)) is None


def test_template_runs_sandboxed(proc):
    evil = tokenizer.ChatTemplate("{{ ''.__class__.__mro__[1].__subclasses__() }}")
    with pytest.raises(Exception):
        evil.render([{"role": "user", "content": "x"}])


def test_template_with_byte_tokenizer_is_refused(proc):
    with pytest.raises(ValueError):
        tokenizer.Processor(proc.cfg, tokenizer.ByteTokenizer(proc.cfg), template_dir=TOKDIR)


def test_paligemma_prompt_ids_equal_hf_processor(kats):
    """BASELINE config 4: PaliGemma's processor has no chat template; it builds "<image> x n <bos> prompt \\n".  The ids of
    HF's own PaliGemmaProcessor (constructible offline) on tests/golden/tokenizer_pg_tiny/ against Processor.chat_ids."""
    pg = kats["paligemma"]
    cfg = engine.preset("tinypg")
    cfg.image_token_id, cfg.bos_id, cfg.eos_ids, cfg.pad_id = pg["image_token_id"], pg["bos_token_id"], (pg["eos_token_id"],), pg["pad_token_id"]
    p = tokenizer.Processor(cfg, tokenizer.HFTokenizer(cfg, os.path.join(GOLD, "tokenizer_pg_tiny")))
    assert p.template is None
    for c in pg["chat"]:
        assert p.chat_ids(c["prompt"], c["image_tokens"]).tolist() == c["input_ids"]
    for d in pg["decode"]:
        assert p.decode(d["ids"], skip_special_tokens=True) == d["skip"]


def test_paligemma_prompt_has_one_bos_when_the_tokenizer_prepends_its_own(kats, tmp_path):
    """A hub Gemma tokenizer.json carries a `<bos> $A` TemplateProcessing post-processor; HF's PaliGemmaProcessor switches
    add_bos_token off before it tokenises and writes the one <bos> itself (paligemma/processing_paligemma.py), so the prompt ids do
    not change.  The golden directory was saved AFTER that switch (empty template), which hid a double <bos> here:
    the same tokenizer with the post-processor put back must give the same ids as HF's processor."""
    import json

    pg = kats["paligemma"]
    src = os.path.join(GOLD, "tokenizer_pg_tiny", "tokenizer.json")
    with open(src) as f:
        tj = json.load(f)
    bos = pg["bos_token_id"]
    bos_tok = next(t["content"] for t in tj["added_tokens"] if t["id"] == bos)
    tj["post_processor"] = {"type": "TemplateProcessing",
                            "single": [{"SpecialToken": {"id": bos_tok, "type_id": 0}}, {"Sequence": {"id": "A", "type_id": 0}}],
                            "pair": [{"SpecialToken": {"id": bos_tok, "type_id": 0}}, {"Sequence": {"id": "A", "type_id": 0}},
                                     {"Sequence": {"id": "B", "type_id": 1}}],
                            "special_tokens": {bos_tok: {"id": bos_tok, "ids": [bos], "tokens": [bos_tok]}}}
    with open(tmp_path / "tokenizer.json", "w") as f:
        json.dump(tj, f)
    cfg = engine.preset("tinypg")
    cfg.image_token_id, cfg.bos_id, cfg.eos_ids, cfg.pad_id = pg["image_token_id"], bos, (pg["eos_token_id"],), pg["pad_token_id"]
    tok = tokenizer.HFTokenizer(cfg, str(tmp_path))
    assert tok.encode("abc")[0] == bos and tok.encode("abc", add_special_tokens=False)[0] != bos   # the post-processor is live
    p = tokenizer.Processor(cfg, tok)
    for c in pg["chat"]:
        ids = p.chat_ids(c["prompt"], c["image_tokens"]).tolist()
        assert ids == c["input_ids"] and ids.count(bos) == 1
