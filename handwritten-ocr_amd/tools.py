"""Drop-in surface of the page-read path: the names `ocr_agent/nodes.py:8-14` imports from `ocr_agent.tools`
(compare_versions, merge_versions, preprocess_image, run_ocr, unload_ocr_model) with the reference's signatures,
prints and singleton semantics, plus the pass-through names of the same module.

    run_ocr(image_path, params=None) -> str      ocr_agent/tools.py:728-771
    _load_ocr_model() -> (model, processor)      ocr_agent/tools.py:683-711   (process-global, lazily built)
    unload_ocr_model() -> None                   ocr_agent/tools.py:714-725

Differences, all behind the same calls: the model is the MI355X read engine (no CPU / other-device path: without the
HIP library or a GPU `run_ocr` raises), and `unload_ocr_model()` keeps the weights resident unless
HWOCR_KEEP_RESIDENT=0 — the reference drops them after every node so Ollama fits on a 48 GB laptop, which on a
288 GB MI355X only buys a checkpoint reload per re-read.  `run_ocr_batch` is the batched entry the reference lacks.

Model selection: HWOCR_MODEL = a checkpoint directory (config.json + *.safetensors + tokenizer.json) — what the
reference's `config.OLMOCR_MODEL` names, downloaded beforehand; it must be set.  The preset names ("qwen2-vl-2b",
"qwen2.5-vl-7b" = "olmocr-2-7b", "qwen2.5-vl-3b", "paligemma-3b", "small", "tiny", "tiny25", "tinypg") build RANDOM-INIT
weights with a byte tokenizer — plausible-looking noise, for benchmarks and tests only — and are refused unless
HWOCR_ALLOW_RANDOM_INIT=1 says that is intended.
"""
from __future__ import annotations

import gc
import os
from pathlib import Path

from PIL import Image

from .compat import config
from .preprocess import preprocess_image  # noqa: F401  (re-exported)
from .text import (_align_to_backbone, _find_differing_segments, _levenshtein_words, cer, compare_versions,  # noqa: F401
                   evaluate, levenshtein, merge_versions, normalize_text, parse_ground_truth, parse_json_response,
                   tier1_metrics, wer)

_ocr_model = None
_ocr_processor = None


def _device() -> str:
    """One process per GPU: the rank's own device (LOCAL_RANK, set by torchrun / bench.py), else the current one."""
    import torch

    from . import shard

    local = os.environ.get("LOCAL_RANK")
    return f"cuda:{shard.local_device_index(int(local))}" if local is not None else f"cuda:{torch.cuda.current_device()}"


def _load_ocr_model():
    global _ocr_model, _ocr_processor
    if _ocr_model is not None:
        return _ocr_model, _ocr_processor
    import torch

    from . import _lib, engine, tokenizer

    if not torch.cuda.is_available():
        raise _lib.HwocrError("run_ocr needs an MI355X: the read engine has no CPU path")
    spec = os.environ.get("HWOCR_MODEL")
    if not spec:
        raise _lib.HwocrError("HWOCR_MODEL is not set: point it at the checkpoint directory of the OCR model "
                              f"(config.json + *.safetensors + tokenizer.json; the reference's default is {config.OLMOCR_MODEL!r})")
    dev = _device()
    # the reference prints its device class ("cuda" on ROCm torch, tools.py:692-699); ranks of a multi-GPU job add their index
    print(f"  [ocr] Loading {spec} on {dev if os.environ.get('LOCAL_RANK') is not None else 'cuda'}...")
    if os.path.isdir(spec):
        if not os.path.exists(os.path.join(spec, "tokenizer.json")):
            raise _lib.HwocrError(f"{spec} has no tokenizer.json: real weights with a stand-in tokenizer would decode to wrong text")
        cfg, sd = engine.load_checkpoint_dir(spec, device=dev)
        tok = tokenizer.HFTokenizer(cfg, spec)
    else:
        if os.environ.get("HWOCR_ALLOW_RANDOM_INIT", "0") in ("", "0"):
            raise _lib.HwocrError(f"HWOCR_MODEL={spec!r} is not a checkpoint directory.  Preset names build random-init weights "
                                  "(noise, for benchmarks and tests): set HWOCR_ALLOW_RANDOM_INIT=1 if that is what you want")
        cfg = engine.preset(spec)
        print("  [ocr] (HWOCR_ALLOW_RANDOM_INIT: random-init weights, byte-level tokenizer — the text is noise)")
        sd = engine.random_state_dict(cfg, seed=int(os.environ.get("HWOCR_SEED", "0")), device=dev)
        tok = tokenizer.ByteTokenizer(cfg, fold_unknown=True)
    cfg.min_pixels, cfg.max_pixels = config.OCR_MIN_PIXELS, config.OCR_MAX_PIXELS
    _ocr_model = engine.ReadEngine(cfg, sd, max_reads=int(os.environ.get("HWOCR_MAX_READS", "256")),
                                   ctx=int(os.environ.get("HWOCR_CTX", "4096")), device=dev,
                                   fp8=os.environ.get("HWOCR_FP8", "0") not in ("", "0"))
    _ocr_processor = tokenizer.Processor(cfg, tok, template_dir=spec if os.path.isdir(spec) else None)
    print("  [ocr] Model loaded.")
    return _ocr_model, _ocr_processor


def _lib_error(msg: str):
    from . import _lib

    return _lib.HwocrError(msg)


_ocr_lanes = None  # pipeline.LanePipeline over _ocr_model: as many lanes as the largest job so far needed (2, or 3)


def plan_lanes(n_reads: int, slots: int, lanes_cfg: int, allow_three: bool = True) -> tuple[int, int]:
    """(lanes, decode slots to use per lane) for a job of n_reads reads on lanes of `slots` decode slots, lanes_cfg = HWOCR_LANES.
    A job that fits one lane runs on one.  Otherwise the reads make `batches` = ceil(n / slots) slot-fills; two lanes take them side
    by side in rounds - unless the job is THREE fills (or an odd multiple of three), which three lanes take in one round instead of
    two lanes in two with the second half empty (a decode step costs nearly the same at 132 rows as at 252, so a half-empty round is
    lost time: a 256-page folder = 768 reads is exactly 3 x 256).  Short jobs (<= 4 rounds) use equal shares of the slots per round
    rather than full rounds and a remainder; long ones keep every slot busy (continuous batching refills them)."""
    batches = -(-n_reads // max(1, slots))
    if batches <= 1 or lanes_cfg <= 1:
        return 1, slots
    lanes = lanes_cfg
    if allow_three and lanes_cfg == 2 and batches % 2 == 1 and batches % 3 == 0:
        lanes = 3
    rounds = -(-batches // lanes)
    per = slots if rounds > 4 else min(slots, -(-n_reads // (rounds * lanes)))
    return lanes, per


def _lanes(n: int | None = None):
    """At least n (default HWOCR_LANES = 2) sets of read slots over the loaded weights, each on its own stream and host thread: a batch
    with more reads than one lane has slots goes through them side by side (pipeline.py: +6 % pages/s on the MI355X, same tokens).
    ONE pipeline is kept and grown when a job needs a third lane (a job with fewer jobs than lanes leaves the last lanes idle).
    None: one lane."""
    global _ocr_lanes
    n = int(os.environ.get("HWOCR_LANES", "2")) if n is None else n
    if n <= 1 or _ocr_model is None:
        return None
    if _ocr_lanes is not None and _ocr_lanes.engines[0] is not _ocr_model:
        _ocr_lanes.close()
        _ocr_lanes = None
    if _ocr_lanes is None or len(_ocr_lanes.engines) < n:
        from . import pipeline

        _ocr_lanes = pipeline.LanePipeline(_ocr_model, lanes=n, reuse=_ocr_lanes.engines[1:] if _ocr_lanes is not None else None)
    return _ocr_lanes


def _close_lanes() -> None:
    global _ocr_lanes
    if _ocr_lanes is not None:
        _ocr_lanes.close()
    _ocr_lanes = None


def unload_ocr_model():
    global _ocr_model, _ocr_processor
    if os.environ.get("HWOCR_KEEP_RESIDENT", "1") == "0" and _ocr_model is not None:
        _close_lanes()
        _ocr_model.close()
        _ocr_model = None
        _ocr_processor = None
        gc.collect()
        import torch

        torch.cuda.empty_cache()
    print("  [ocr] Model unloaded, memory freed.")


def run_ocr_batch_tokens(images: list, params: dict | None = None, on_done=None) -> list[list[int]]:
    """Read many (already preprocessed) pages in one engine batch; the generated token ids of every read (what the
    multi-GPU driver gathers to rank 0, shard.gather_token_streams).  `images`: paths, PIL images, or uint8 [H][W][3] device
    tensors already at the tower's resolution (gpupre.StrategyPages).  `on_done(i, tokens)`: called as soon as read i has stopped
    (from the thread that drives its lane - two lanes call it concurrently)."""
    params = params or {}
    model, processor = _load_ocr_model()
    prompt = params.get("prompt", config.OCR_PROMPT)
    max_new = params.get("max_new_tokens", config.OCR_MAX_NEW_TOKENS)
    min_new = params.get("min_new_tokens", 0)
    pages, prompts = [], []
    if hasattr(images, "image_tokens"):   # batch._LazyDeviceReads: a read's page is made when a lane admits it; its size is known now
        pages, by_n = images, {}
        for i in range(len(images)):
            n = images.image_tokens(i)
            if n not in by_n:
                by_n[n] = processor.chat_ids(prompt, n)
            prompts.append(by_n[n])
        images = []
    for im in images:
        if hasattr(im, "data_ptr"):  # torch tensor resident in HBM
            pages.append(im)
            prompts.append(processor.chat_ids(prompt, processor.image_tokens(im)))
            continue
        img = im if isinstance(im, Image.Image) else Image.open(im)
        page, ids = processor.prepare(img, prompt)
        pages.append(page)
        prompts.append(ids)
    # continuous batching: reads stop at different lengths (EOS), freed decode slots take the next read
    # (read_ids: the caller's numbering of the reads - it keys the sampling RNG, so that a sampled read does not depend on the shard)
    rp = params.get("repetition_penalty")  # None: the checkpoint's default
    ids = params.get("read_ids")
    lanes_cfg = int(os.environ.get("HWOCR_LANES", "2"))
    n_lanes, per_lane = plan_lanes(len(pages), model.max_reads, lanes_cfg)
    try:
        pipe = _lanes(n_lanes) if n_lanes > 1 else None
    except RuntimeError as e:   # a third lane's KV cache did not fit beside the other two (a big model at a long context): two lanes
        import torch

        if n_lanes <= lanes_cfg or not isinstance(e, torch.cuda.OutOfMemoryError):
            raise
        torch.cuda.empty_cache()
        n_lanes, per_lane = plan_lanes(len(pages), model.max_reads, lanes_cfg, allow_three=False)
        pipe = _lanes(n_lanes) if n_lanes > 1 else None
    if pipe is None:
        return model.generate_stream(pages, prompts, max_new=max_new, min_new=min_new, repetition_penalty=rp, read_ids=ids,
                                     on_done=on_done)
    # more reads than one lane has slots: the lanes work side by side through ONE queue of waiting reads (engine.ReadSource), each
    # taking the next ones whenever its slots free up (a read's tokens do not depend on its lane: the RNG of a sampled read is keyed
    # by the caller's read number)
    from . import engine as _engine

    ids = list(range(len(pages))) if ids is None else list(ids)
    src = _engine.ReadSource(len(pages))
    jobs = [(lambda e, hooks: e.generate_stream(pages, prompts, max_new=max_new, min_new=min_new, repetition_penalty=rp, read_ids=ids,
                                                on_done=on_done, source=src, max_slots=per_lane)) for _ in range(n_lanes)]
    parts = pipe.run(jobs)
    out = [next((part[i] for part in parts if part[i] is not None), None) for i in range(len(pages))]
    if any(t is None for t in out):
        raise _lib_error("a read was taken by no lane")
    return out


def decode_tokens(streams: list) -> list[str]:
    """Generated ids -> text as the reference does it (tools.py:767-769: new tokens only, skip_special_tokens=True)."""
    _, processor = _load_ocr_model()
    return [processor.decode(toks, skip_special_tokens=True) for toks in streams]


def run_ocr_batch(images: list, params: dict | None = None) -> list[str]:
    return decode_tokens(run_ocr_batch_tokens(images, params))


def run_ocr(image_path: str, params: dict | None = None) -> str:
    print(f"  [ocr] Running OCR on {Path(image_path).name}...")
    result = run_ocr_batch([str(image_path)], params)[0]
    print(f"  [ocr] Done ({len(result)} chars)")
    return result


_PATCHED = ("compare_versions", "merge_versions", "preprocess_image", "run_ocr", "unload_ocr_model", "_load_ocr_model",
            "levenshtein", "_levenshtein_words", "cer", "wer", "tier1_metrics", "normalize_text")


def install() -> None:
    """Bind this module's functions into an installed reference package: `ocr_agent.tools` always, and `ocr_agent.nodes`
    too when it has already been imported (it copies the names at import time, nodes.py:8-14)."""
    import importlib
    import sys

    ref_tools = importlib.import_module("ocr_agent.tools")
    me = sys.modules[__name__]
    for name in _PATCHED:
        setattr(ref_tools, name, getattr(me, name))
    nodes = sys.modules.get("ocr_agent.nodes")
    if nodes is not None:
        for name in ("compare_versions", "merge_versions", "preprocess_image", "run_ocr", "unload_ocr_model"):
            setattr(nodes, name, getattr(me, name))
