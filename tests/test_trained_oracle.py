"""The accuracy bar of the path — output CER within 0.5 % of the reference's (BASELINE.json; metric: `cer`, ocr_agent/tools.py:103-118,
over the text `generate` returns, tools.py:764-769) — on the CPU side: the fixtures tests/golden/trained_* hold what the real HF
classes transcribe from 8 synthetic pages with a briefly TRAINED tiny checkpoint (decisive greedy choices: top-1 / top-2 margin > 1
on every step but one, unlike the random-init goldens whose free-running streams mean nothing), read the way the reference reads its
model (tools/make_goldens.py::make_trained).  Here: the oracle free-running on the same inputs must give HF's token streams exactly,
and the host half of the drop-in (chat template, placeholder expansion, tokenizer, image processor grid, detokeniser) must give HF's
prompt ids and HF's text.  The GPU half is tests/test_trained_gpu.py."""
import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import load_file

from handwritten_ocr_amd import engine, imageproc, tokenizer
from handwritten_ocr_amd.compat import config
from oracle import image_ref
from oracle.qwen2vl_ref import Qwen2VLRef, RefConfig
from tests._golden import TRAINED_FAMILIES as FAMILIES, mean_cer, trained_dir, trained_meta, trained_page


def _ref_config(cfg: engine.ModelConfig) -> RefConfig:
    tower = dict(embed_dim=cfg.embed_dim, mlp_ratio=cfg.mlp_ratio) if cfg.family == "qwen2_vl" else \
        dict(embed_dim=cfg.embed_dim, family=cfg.family, vit_inter=cfg.vit_inter, window_size=cfg.window_size, fullatt=tuple(cfg.fullatt))
    return RefConfig(depth=cfg.depth, num_heads=cfg.num_heads, patch_size=cfg.patch_size, merge=cfg.merge, tps=cfg.tps,
                     hidden=cfg.hidden, layers=cfg.layers, q_heads=cfg.q_heads, kv_heads=cfg.kv_heads, inter=cfg.inter,
                     vocab=cfg.vocab, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), eps=cfg.eps,
                     image_token_id=cfg.image_token_id, vision_start_id=cfg.vision_start_id, vision_end_id=cfg.vision_end_id,
                     tie=True, eos_ids=tuple(cfg.eos_ids), pad_id=cfg.pad_id, **tower)


@pytest.mark.parametrize("family", FAMILIES)
def test_checkpoint_dir_loads_with_its_generation_defaults(family):
    meta = trained_meta(family)
    cfg, sd = engine.load_checkpoint_dir(trained_dir(family), device="cpu")
    cfg.validate()
    assert cfg.family == family
    assert list(cfg.eos_ids) == meta["eos_token_id"] and cfg.pad_id == meta["pad_token_id"]
    assert not cfg.do_sample and cfg.repetition_penalty == 1.0
    assert all(v.dtype == torch.bfloat16 for v in sd.values())


def _siglip_pixel_values(img: Image.Image, size: int) -> torch.Tensor:
    """HF SiglipImageProcessorPil restated (resize BICUBIC, x 1/255 in float64 cast to float32, (x - 0.5) / 0.5 in float32): [3, S, S]."""
    arr = np.asarray(img.convert("RGB").resize((size, size), resample=Image.BICUBIC)).transpose(2, 0, 1)
    x = (arr.astype(np.float64) * (1 / 255)).astype(np.float32)
    return torch.from_numpy((x - np.float32(0.5)) / np.float32(0.5))


@pytest.mark.parametrize("family", FAMILIES)
def test_host_half_of_run_ocr_gives_hf_prompt_ids_and_text(family):
    """What tools.run_ocr does around the engine, on the checkpoint directory's own files: prompt ids ≡ HF's
    apply_chat_template + expansion + tokenizer, page grid ≡ HF's image processor, decode(HF's new tokens) ≡ HF's text."""
    meta = trained_meta(family)
    cfg, _ = engine.load_checkpoint_dir(trained_dir(family), device="cpu")
    cfg.min_pixels, cfg.max_pixels = config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS   # tools._load_ocr_model (tools.py:700-704)
    if family != "paligemma":
        assert (cfg.min_pixels, cfg.max_pixels) == (meta["min_pixels"], meta["max_pixels"])
    proc = tokenizer.Processor(cfg, tokenizer.HFTokenizer(cfg, trained_dir(family)), template_dir=trained_dir(family))
    assert meta["prompt"] == config.OCR_PROMPT
    for case in meta["cases"]:
        page, ids = proc.prepare(Image.fromarray(trained_page(case), "RGB"), meta["prompt"])
        if family == "paligemma":
            assert page.shape[:2] == (meta["image_size"], meta["image_size"])
        else:
            assert [1, page.shape[0] // cfg.patch_size, page.shape[1] // cfg.patch_size] == case["grid_thw"]
        assert ids.tolist() == case["input_ids"]
        assert proc.decode(case["hf_tokens"], skip_special_tokens=True) == case["hf_text"]


@pytest.mark.parametrize("family", FAMILIES)
def test_oracle_free_running_stream_is_hfs(family):
    """bf16 oracle, free-running greedy with the checkpoint's EOS set: the same tokens as HF's generate on every page (the streams
    that stop on EOS stop at the same step, the others run to the 128-token budget), hence CER 0 against HF's text."""
    meta = trained_meta(family)
    cfg, sd = engine.load_checkpoint_dir(trained_dir(family), device="cpu")
    if family == "paligemma":
        from oracle.paligemma_ref import PaliGemmaRef, PaliRefConfig

        ref = PaliGemmaRef(PaliRefConfig(
            v_layers=cfg.depth, v_hidden=cfg.embed_dim, v_heads=cfg.num_heads, v_inter=cfg.vit_inter, patch_size=cfg.patch_size,
            image_size=cfg.image_size, hidden=cfg.hidden, layers=cfg.layers, q_heads=cfg.q_heads, kv_heads=cfg.kv_heads, head_dim=cfg.head_dim,
            inter=cfg.inter, vocab=cfg.vocab, rope_theta=cfg.rope_theta, image_token_id=cfg.image_token_id, eos_ids=tuple(cfg.eos_ids),
            pad_id=cfg.pad_id), sd)
    else:
        ref = Qwen2VLRef(_ref_config(cfg), sd)
    proc = tokenizer.Processor(cfg, tokenizer.HFTokenizer(cfg, trained_dir(family)), template_dir=trained_dir(family))
    n = meta["max_new_tokens"]
    texts = []
    for case in meta["cases"]:
        img = Image.fromarray(trained_page(case), "RGB")
        if family == "paligemma":
            toks, logits = ref.generate(torch.tensor(case["input_ids"]), _siglip_pixel_values(img, meta["image_size"]).to(torch.bfloat16), max_new=n)
        else:
            pv, grid = image_ref.pixel_values(img, meta["min_pixels"], meta["max_pixels"])
            assert list(grid) == case["grid_thw"]
            toks, logits = ref.generate(torch.tensor(case["input_ids"]), torch.from_numpy(pv), [grid], max_new=n)
        assert toks == case["hf_tokens"], (case["page_seed"], toks[:8], case["hf_tokens"][:8])
        # the fixture's own claim: decisive steps
        top2 = logits.float().topk(2, -1).values
        margins = (top2[:, 0] - top2[:, 1]).numpy()
        assert np.allclose(margins, np.asarray(case["margins"][: len(margins)]), atol=0.26), "margins are those HF recorded (bf16 logits: 1-2 ulps of ~16)"
        texts.append(proc.decode(toks, skip_special_tokens=True))
    assert mean_cer([c["hf_text"] for c in meta["cases"]], texts) == 0.0
    stopped = [c["stopped_on_eos"] for c in meta["cases"]]
    assert any(stopped) and not all(stopped), "the fixture holds both kinds of stream: stopped by EOS, cut by the budget"
    assert len({c["hf_text"] for c in meta["cases"]}) == len(meta["cases"]), "every page reads differently: the image decides"
    assert meta["decisive_fraction_margin_gt_1"] >= 0.95
