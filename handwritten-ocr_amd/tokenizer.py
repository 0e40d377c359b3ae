"""Prompt construction and detokenisation around the read engine.

The reference gets both from the checkpoint's processor files: `processor.apply_chat_template(messages,
add_generation_prompt=True, tokenize=True, ...)` and `processor.decode(new_ids, skip_special_tokens=True)`
(ocr_agent/tools.py:744-769).  Restated here without `transformers`:

  ChatTemplate    the checkpoint's jinja chat template (chat_template.jinja / chat_template.json / processor_config.json /
                  tokenizer_config.json, the places HF's processor and tokenizer look), rendered in the same sandboxed
                  environment HF builds (utils/chat_template_utils.py:424-495: trim_blocks, lstrip_blocks, loop controls,
                  `raise_exception`, `strftime_now`, non-escaping `tojson`) on the message list run_ocr sends: one user turn,
                  an image item then the prompt text.
  HFTokenizer     `tokenizer.json` of the checkpoint through the `tokenizers` library (what PreTrainedTokenizerFast wraps).
  Processor       render -> expand the single image placeholder to the page's token count (HF Qwen2VLProcessor
                  .replace_image_token, processing_qwen2_vl.py:58-61) -> encode.
Pinned by tests/golden/tokenizer_kats.json (outputs of HF's own tokenizer / template renderer / PIL image processor).

Without tokenizer files (random-init presets: benchmarks and tests) ByteTokenizer stands in: 256 byte tokens + the model
family's special-token ids, with the Qwen2-VL layout built in: system turn, user turn = <|vision_start|> image placeholders
<|vision_end|> followed by the prompt text, then the assistant generation prefix.
"""
from __future__ import annotations

import json
import os

import numpy as np

SYSTEM_TEXT = "You are a helpful assistant."


class ChatTemplate:
    """A checkpoint's chat template, rendered as HF renders it."""

    FILES = ("chat_template.jinja", "chat_template.json", "processor_config.json", "tokenizer_config.json")

    def __init__(self, source: str, special_tokens: dict | None = None):
        import jinja2
        import jinja2.ext
        from jinja2.sandbox import ImmutableSandboxedEnvironment

        def raise_exception(message):
            raise jinja2.exceptions.TemplateError(message)

        def tojson(x, ensure_ascii=False, indent=None, separators=None, sort_keys=False):
            return json.dumps(x, ensure_ascii=ensure_ascii, indent=indent, separators=separators, sort_keys=sort_keys)

        def strftime_now(fmt):
            from datetime import datetime

            return datetime.now().strftime(fmt)

        env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True, extensions=[jinja2.ext.loopcontrols])
        env.filters["tojson"] = tojson
        env.globals["raise_exception"] = raise_exception
        env.globals["strftime_now"] = strftime_now
        self.source = source
        self.special_tokens = dict(special_tokens or {})
        self._tpl = env.from_string(source)

    @classmethod
    def from_dir(cls, path: str) -> "ChatTemplate | None":
        """The template a processor loaded from `path` would use (processor files first, then the tokenizer's), or None."""
        special = {}
        cfg_path = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(cfg_path):
            with open(cfg_path, encoding="utf-8") as f:
                tc = json.load(f)
            for k in ("bos_token", "eos_token", "unk_token", "sep_token", "pad_token", "cls_token", "mask_token"):
                v = tc.get(k)
                if isinstance(v, dict):
                    v = v.get("content")
                if isinstance(v, str):
                    special[k] = v
        for fn in cls.FILES:
            fp = os.path.join(path, fn)
            if not os.path.exists(fp):
                continue
            if fn.endswith(".jinja"):
                with open(fp, encoding="utf-8") as f:
                    return cls(f.read(), special)
            with open(fp, encoding="utf-8") as f:
                tpl = json.load(f).get("chat_template")
            if isinstance(tpl, list):  # named templates: [{"name": "default", "template": ...}, ...]
                tpl = next((t.get("template") for t in tpl if t.get("name") == "default"), None)
            if isinstance(tpl, str):
                return cls(tpl, special)
        return None

    def render(self, messages: list, add_generation_prompt: bool = True, **kwargs) -> str:
        return self._tpl.render(messages=messages, tools=None, documents=None, add_generation_prompt=add_generation_prompt,
                                **self.special_tokens, **kwargs)


class ByteTokenizer:
    def __init__(self, cfg, fold_unknown: bool = False):
        self.cfg = cfg
        self.fold_unknown = fold_unknown  # random-init models emit arbitrary ids: fold them onto bytes for synthetic text
        self.special = {cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.im_start_id, cfg.im_end_id,
                        cfg.pad_id, *cfg.eos_ids}

    def encode(self, text: str, add_special_tokens: bool = True) -> list[int]:
        return list(text.encode("utf-8"))  # (no post-processor: the flag changes nothing)

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
    """`tokenizer.json` of a checkpoint directory.  encode = PreTrainedTokenizerFast.__call__(text)["input_ids"] (the
    post-processor's special tokens included, none for the Qwen byte-level BPE); decode = PreTrainedTokenizerFast.decode with
    its defaults (clean_up_tokenization_spaces off)."""

    def __init__(self, cfg, path: str):
        from tokenizers import Tokenizer

        self.cfg = cfg
        self.tok = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))

    def encode(self, text: str, add_special_tokens: bool = True) -> list[int]:
        """add_special_tokens: run tokenizer.json's post-processor (what PreTrainedTokenizerFast.__call__ does on the whole rendered
        prompt); False for PIECES of a prompt that the caller assembles around explicit special ids (a `<bos> $A` template —
        hub Gemma tokenizers — would otherwise put a second <bos> in front of every piece)."""
        return self.tok.encode(text, add_special_tokens=add_special_tokens).ids

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        return self.tok.decode([int(t) for t in ids], skip_special_tokens=skip_special_tokens)

    def token_of(self, token_id: int) -> str | None:
        return self.tok.id_to_token(int(token_id))


class Processor:
    """Page + prompt text -> (tower-resolution pixels, prompt ids); ids -> text."""

    def __init__(self, cfg, tokenizer, template_dir: str | None = None):
        """template_dir: the checkpoint directory; its chat template (if any) replaces the built-in layout."""
        self.cfg = cfg
        self.tokenizer = tokenizer
        self.template = ChatTemplate.from_dir(template_dir) if template_dir else None
        if self.template is not None and not hasattr(tokenizer, "token_of"):
            raise ValueError("a chat template needs the checkpoint's tokenizer (tokenizer.json): the byte tokenizer cannot encode its special tokens")

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

    def chat_text(self, prompt: str, image_ref: str = "page") -> str:
        """The checkpoint's chat template on run_ocr's message list (tools.py:744-753); one image placeholder inside."""
        messages = [{"role": "user", "content": [{"type": "image", "url": image_ref}, {"type": "text", "text": prompt}]}]
        return self.template.render(messages, add_generation_prompt=True)

    def chat_ids(self, prompt: str, n_image_tokens: int) -> np.ndarray:
        c, enc = self.cfg, self.tokenizer.encode
        if self.template is not None:
            text = self.chat_text(prompt)
            pad = self.tokenizer.token_of(c.image_token_id)
            if pad is None or text.count(pad) != 1:
                raise ValueError(f"the chat template must yield exactly one {pad!r} for one image, got {0 if pad is None else text.count(pad)}")
            ids = np.asarray(enc(text.replace(pad, pad * n_image_tokens)), dtype=np.int32)  # processing_qwen2_vl.py:58-61
            if int((ids == c.image_token_id).sum()) != n_image_tokens:
                raise ValueError("tokenizer.json does not keep the image placeholder as one token")
            return ids
        if c.family == "paligemma":
            # "<image>" x n + "<bos>" + prompt + "\n" tokenised as one string whose specials split it (HF
            # paligemma/processing_paligemma.py build_string_from_input): the text part is encoded WITH its newline, and WITHOUT
            # the tokenizer's own specials — PaliGemmaProcessor switches add_bos_token off and writes the one <bos> itself
            return np.asarray([c.image_token_id] * n_image_tokens + [c.bos_id] + enc(prompt + "\n", add_special_tokens=False),
                              dtype=np.int32)
        enc = lambda t: self.tokenizer.encode(t, add_special_tokens=False)  # noqa: E731  (pieces around explicit special ids)
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
