"""Prompt construction around the engine (handwritten_ocr_amd/tokenizer.py): chat layout per family, placeholder counts that
match what the engine will splice, byte-tokenizer round trip.  CPU only."""
import numpy as np
from PIL import Image

from handwritten_ocr_amd import engine, imageproc, tokenizer


def test_byte_tokenizer_round_trip_and_specials():
    cfg = engine.preset("tiny")
    tok = tokenizer.ByteTokenizer(cfg)
    text = "Grüße, “page” — naïve café\n2nd line"
    ids = tok.encode(text)
    assert all(0 <= t < 256 for t in ids) and tok.decode(ids) == text
    assert tok.decode(ids + [cfg.eos_ids[0], cfg.pad_id]) == text            # specials dropped by default
    assert f"<|{cfg.eos_ids[0]}|>" in tok.decode([65, cfg.eos_ids[0]], skip_special_tokens=False)
    folded = tokenizer.ByteTokenizer(cfg, fold_unknown=True).decode([300, 301])
    assert len(folded) == 2 and folded.isprintable()


def test_qwen_chat_layout_and_placeholder_count():
    for preset in ("tiny", "tiny25"):
        cfg = engine.preset(preset)
        proc = tokenizer.Processor(cfg, tokenizer.ByteTokenizer(cfg))
        page, ids = proc.prepare(Image.new("RGB", (200, 150), "white"), "Read it.")
        assert page.dtype == np.uint8 and page.shape[0] % 28 == 0 and page.shape[1] % 28 == 0
        n_img = (page.shape[0] // 14) * (page.shape[1] // 14) // 4
        img = np.nonzero(ids == cfg.image_token_id)[0]
        assert len(img) == n_img == proc.image_tokens(page)
        assert (np.diff(img) == 1).all(), "one contiguous run of placeholders"
        assert ids[img[0] - 1] == cfg.vision_start_id and ids[img[-1] + 1] == cfg.vision_end_id
        assert ids[0] == cfg.im_start_id and list(ids).count(cfg.im_start_id) == 3 and list(ids).count(cfg.im_end_id) == 2
        # the positions the engine derives for this prompt: text runs count up, the image run advances by max(h, w) / merge
        pos, delta = imageproc.mrope_positions(ids, cfg.image_token_id, [(1, page.shape[0] // 14, page.shape[1] // 14)], cfg.merge)
        assert pos.shape == (3, len(ids)) and int(pos.max()) + 1 - len(ids) == delta < 0


def test_paligemma_prompt_layout():
    cfg = engine.preset("tinypg")
    proc = tokenizer.Processor(cfg, tokenizer.ByteTokenizer(cfg))
    page, ids = proc.prepare(Image.new("RGB", (300, 120), "white"), "ocr")
    assert page.shape == (cfg.image_size, cfg.image_size, 3)                 # plain square resize, aspect not kept
    n = (cfg.image_size // cfg.patch_size) ** 2
    assert ids[:n].tolist() == [cfg.image_token_id] * n and ids[n] == cfg.bos_id
    assert bytes(ids[n + 1:].tolist()) == b"ocr\n"
