"""Prompt construction and detokenisation around the read engine.

The reference gets both from the checkpoint's processor files (`processor.apply_chat_template` / `processor.decode`,
ocr_agent/tools.py:756-769).  No tokenizer file exists offline, so two back ends are provided:
  ByteTokenizer   256 byte tokens + the model family's special-token ids; lets random-init models round-trip text.
  HFTokenizer     `tokenizer.json` of a real checkpoint directory through the `tokenizers` library.
Both render the Qwen2-VL chat layout: system turn, user turn = <|vision_start|> image placeholders <|vision_end|>
followed by the prompt text, then the assistant generation prefix.
"""
from __future__ import annotations

import os

import numpy as np

SYSTEM_TEXT = "You are a helpful assistant."


class ByteTokenizer:
    def __init__(self, cfg, fold_unknown: bool = False):
        self.cfg = cfg
        self.fold_unknown = fold_unknown  # random-init models emit arbitrary ids: fold them onto bytes for synthetic text
        self.special = {cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.im_start_id, cfg.im_end_id,
                        cfg.pad_id, *cfg.eos_ids}

    def encode(self, text: str) -> list[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        out = bytearray()
        for t in ids:
            t = int(t)
            if t in self.special:
                if not skip_special_tokens:
                    out += f"<|{t}|>".encode()
            elif t < 256:
                out.append(t)
            elif self.fold_unknown:
                out.append(32 + t % 95)  # printable ASCII
        return out.decode("utf-8", errors="replace")


class HFTokenizer:
    def __init__(self, cfg, path: str):
        from tokenizers import Tokenizer

        self.cfg = cfg
        self.tok = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))

    def encode(self, text: str) -> list[int]:
        return self.tok.encode(text, add_special_tokens=False).ids

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        return self.tok.decode([int(t) for t in ids], skip_special_tokens=skip_special_tokens)


class Processor:
    """Page + prompt text -> (tower-resolution pixels, prompt ids); ids -> text."""

    def __init__(self, cfg, tokenizer):
        self.cfg = cfg
        self.tokenizer = tokenizer

    def target_hw(self, height: int, width: int) -> tuple[int, int]:
        """Resolution `prepare` resizes a page of this size to (what the device preprocessing must produce)."""
        from . import imageproc

        c = self.cfg
        if c.family == "paligemma":
            return c.image_size, c.image_size
        return imageproc.smart_resize(height, width, c.patch_size * c.merge, c.min_pixels, c.max_pixels)

    def image_tokens(self, page: np.ndarray) -> int:
        c = self.cfg
        return (page.shape[0] // c.patch_size) * (page.shape[1] // c.patch_size) // c.merge ** 2

    def chat_ids(self, prompt: str, n_image_tokens: int) -> np.ndarray:
        c, enc = self.cfg, self.tokenizer.encode
        if c.family == "paligemma":  # <image> x n, <bos>, prompt, newline (HF paligemma/processing_paligemma.py build_string_from_input)
            return np.asarray([c.image_token_id] * n_image_tokens + [c.bos_id] + enc(prompt) + enc("\n"), dtype=np.int32)
        ids = ([c.im_start_id] + enc("system\n" + SYSTEM_TEXT) + [c.im_end_id] + enc("\n")
               + [c.im_start_id] + enc("user\n") + [c.vision_start_id] + [c.image_token_id] * n_image_tokens
               + [c.vision_end_id] + enc(prompt) + [c.im_end_id] + enc("\n") + [c.im_start_id] + enc("assistant\n"))
        return np.asarray(ids, dtype=np.int32)

    def prepare(self, img, prompt: str):
        from . import imageproc

        c = self.cfg
        if c.family == "paligemma":
            page = imageproc.prepare_square(img, c.image_size)
            return page, self.chat_ids(prompt, self.image_tokens(page))
        page = imageproc.prepare_page(img, c.patch_size, c.merge, c.min_pixels, c.max_pixels)
        return page, self.chat_ids(prompt, self.image_tokens(page))

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        return self.tokenizer.decode(ids, skip_special_tokens=skip_special_tokens)
