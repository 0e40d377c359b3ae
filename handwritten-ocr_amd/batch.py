"""Batch-folder driver with cross-page batching.

The reference walks a folder one page at a time and, inside a page, one read at a time
(ocr_agent/transcribe.py:185-210 -> graph.invoke per page -> nodes.py:86-110), so its engine never sees more than one
sequence.  Here every page's three strategy reads are prepared up front, all reads of all pages go through the engine in
batches of `max_reads`, and only then is each page's `initial_ocr` node replayed — by the same node code, with
`preprocess_image` / `run_ocr` answering from the finished batch — so the candidates, trace events (order included) and
merged text are exactly what the serial path produces.  The third (tie-breaker) read is speculative: it is computed for
every page and simply not consumed when reads 1 and 2 agree (nodes.py:109).

Per page the reference's four output files are written with the same names and formats
(transcribe.py:76-101, trace.py:56-70): `<stem>_transcription.txt`, `_trace.json`, `_trace_summary.txt`, `_eval.json`.
The critic / editor / re-OCR loop needs the LLM agents, which are out of scope: pass callables through `agents=` to run
the full graph (`compat.nodes.run_graph`); without them a page stops after `initial_ocr` with status "initial_ocr".
With agents, every DISTINCT preprocessing strategy of every page is read in the same batched pass: a later
`reocr` node (nodes.py:239-302: next unused strategy, one more read, after a model reload in the reference) is then
answered from that pass — same node code, same text, no second trip through the engine.

Multi-GPU: one process per GPU (`torchrun`), pages dealt round-robin (`shard.shard`), every rank writes its own pages'
files; token streams are additionally gathered to rank 0 by `tools.run_ocr_batch` callers that need them (bench.py).

CLI:  python -m handwritten_ocr_amd.batch <folder-or-image> [--output-dir D] [--ground-truth-dir G] [--max-new-tokens N]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from PIL import Image

from . import preprocess, shard, tools
from .compat import config, nodes
from .compat.state import new_state

IMAGE_EXTENSIONS = {".png", ".jpg", ".jpeg", ".bmp", ".tiff", ".tif", ".webp"}


def list_images(folder: Path) -> list[Path]:
    return sorted(f for f in Path(folder).iterdir() if f.suffix.lower() in IMAGE_EXTENSIONS)


def _speculative_strategies(strategies: list, every: bool = False) -> list:
    """The (at most three) distinct strategies node_initial_ocr can touch, in its order (nodes.py:86-110, dedup :36-39);
    `every`: all distinct strategies, i.e. also those node_reocr would pick next (nodes.py:246-251)."""
    out, seen = [], set()
    for s in ((strategies if every else strategies[:3]) if strategies else ["original"]):
        label = nodes._strategy_label(s)
        if label not in seen:
            seen.add(label)
            out.append(s)
    return out


def _replay_readers(strategies: list, texts: list, fallback=None):
    """(preprocess_image, run_ocr) stand-ins that answer one page's reads from the finished batch, printing what the
    real ones print; strategies outside the batch go to `fallback` = the real (preprocess_image, run_ocr) pair."""
    labels = [nodes._strategy_label(s) for s in strategies]
    by_token: dict = {}

    def replay_preprocess(image_path, strategy):
        label = nodes._strategy_label(strategy)
        if label not in labels:
            if fallback is None:
                raise KeyError(f"strategy {label!r} was not part of the batched pass")
            return fallback[0](image_path, strategy)
        token = f"{image_path}#{label}"
        by_token[token] = texts[labels.index(label)]
        if preprocess.steps_of(strategy) not in (["original"], []):
            print(f"  [preprocess] Applying {preprocess.label_of(strategy)}...")
        return token

    def replay_run_ocr(token, params=None):
        if token not in by_token:
            return fallback[1](token, params)
        print(f"  [ocr] Running OCR on {Path(token.split('#')[0]).name}...")
        print(f"  [ocr] Done ({len(by_token[token])} chars)")
        return by_token[token]

    return replay_preprocess, replay_run_ocr


def initial_ocr_batched(image_paths: list, params: dict | None = None, cfg=config, workers: int = 8,
                        speculate_reocr: bool = False, reads_out: list | None = None) -> list[dict]:
    """States after `initial_ocr` for every page, computed with ONE batched engine pass over all reads.
    `speculate_reocr`: also read the strategies a later `reocr` node would use; `reads_out` (a list) receives, per page,
    (strategies, texts) of everything that was read."""
    strategies = _speculative_strategies(list(cfg.PREPROCESSING_STRATEGIES), every=speculate_reocr)

    # HWOCR_GPU_PREPROCESS=1: the strategy chains and the processor's resize run on the device, bit-identical
    # to the host path (gpupre.py); a page is decoded and uploaded once for all its reads.  Pages that are not plain RGB and
    # chains the device path does not cover (OpenCV present, a transform after binarize) keep the host path.
    gpu_pages = None
    if os.environ.get("HWOCR_GPU_PREPROCESS", "0") not in ("", "0"):
        from . import gpupre

        if all(gpupre.supported(s) for s in strategies):
            model, processor = tools._load_ocr_model()
            gpu_pages = (gpupre.StrategyPages(model.dev), processor)

    def prepare(path):
        img = Image.open(path)
        img.load()
        if gpu_pages is not None and img.mode == "RGB":
            return np.asarray(img)
        return [preprocess.apply_strategy(img, s, quiet=True) for s in strategies]

    with ThreadPoolExecutor(max_workers=workers) as pool:
        prepared = list(pool.map(prepare, image_paths))
    if gpu_pages is not None:
        sp, processor = gpu_pages
        prepared = [sp.pages(p, strategies, processor.target_hw(p.shape[0], p.shape[1])) if isinstance(p, np.ndarray) else p
                    for p in prepared]
    flat = [im for page in prepared for im in page]
    texts = tools.run_ocr_batch(flat, params)
    states = []
    k = len(strategies)
    for p, path in enumerate(image_paths):
        page_texts = texts[p * k: (p + 1) * k]
        fake_preprocess, fake_run_ocr = _replay_readers(strategies, page_texts)
        if reads_out is not None:
            reads_out.append((strategies, page_texts))
        state = new_state(str(path), cfg)
        saved = (nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model)
        nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model = fake_preprocess, fake_run_ocr, (lambda: None)
        try:
            state.update(nodes.node_initial_ocr(state))
        finally:
            nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model = saved
        states.append(state)
    return states


def _fmt_elapsed(seconds: float) -> str:
    m, s = divmod(int(seconds), 60)
    return f"{m:02d}:{s:02d}"


def write_outputs(state: dict, output_dir: Path, ground_truth_path: Path | None = None) -> Path:
    """The reference's four per-page files (transcribe.py:76-101; trace.py:56-70)."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    name = Path(state["image_path"]).stem
    out = output_dir / f"{name}_transcription.txt"
    out.write_text(state["current_best"], encoding="utf-8")
    with open(output_dir / f"{name}_trace.json", "w", encoding="utf-8") as f:
        json.dump(state["trace_events"], f, indent=2, ensure_ascii=False)
    lines = [f"[{_fmt_elapsed(e['elapsed_seconds'])}] {e['output_summary']}" for e in state["trace_events"]]
    (output_dir / f"{name}_trace_summary.txt").write_text("\n".join(lines) + "\n", encoding="utf-8")
    gt = tools.parse_ground_truth(ground_truth_path) if ground_truth_path else None
    result = tools.evaluate(state["current_best"], ground_truth=gt)
    result["pipeline_status"] = state["status"]
    result["iterations"] = state["iteration"]
    result["final_confidence"] = state["current_score"]
    with open(output_dir / f"{name}_eval.json", "w", encoding="utf-8") as f:
        json.dump(result, f, indent=2, ensure_ascii=False)
    return out


def transcribe_folder(images: list, output_dir: Path, ground_truth_dir: Path | None = None, params: dict | None = None,
                      agents: dict | None = None, quiet: bool = False) -> list[Path]:
    """Batched `initial_ocr` for this rank's share of `images`, then (with agents) the rest of the graph per page."""
    rank, _, world = shard.init_from_env()
    mine = shard.shard([Path(p) for p in images], rank, world)
    sink = io.StringIO() if quiet else None
    reads: list = []
    with contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext():
        states = initial_ocr_batched([str(p) for p in mine], params, speculate_reocr=bool(agents), reads_out=reads)
    outs = []
    for state, (strategies, page_texts) in zip(states, reads):
        if agents:
            nodes.run_critic, nodes.run_editor = agents["critic"], agents["editor"]
            nodes.run_arbitrator = agents.get("arbitrator")
            # re-OCR rounds read the strategies that were prefetched in the batched pass; anything else goes to the engine
            saved = (nodes.preprocess_image, nodes.run_ocr)
            nodes.preprocess_image, nodes.run_ocr = _replay_readers(strategies, page_texts, fallback=saved)
            try:
                with contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext():
                    state = nodes.run_graph_after_initial(state)
            finally:
                nodes.preprocess_image, nodes.run_ocr = saved
        else:
            state["status"], state["reason"] = "initial_ocr", "no_agents"
        gt = None
        if ground_truth_dir:
            for ext in (".md", ".txt"):
                cand = Path(ground_truth_dir) / f"{Path(state['image_path']).stem}{ext}"
                if cand.exists():
                    gt = cand
                    break
        outs.append(write_outputs(state, output_dir, gt))
    return outs


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Batched handwritten-page transcription on MI355X")
    ap.add_argument("input", type=Path)
    ap.add_argument("--output-dir", type=Path, default=None)
    ap.add_argument("--ground-truth-dir", type=Path, default=None)
    ap.add_argument("--max-new-tokens", type=int, default=None)
    args = ap.parse_args(argv)
    src = args.input.resolve()
    if not src.exists():
        print(f"Error: {src} does not exist", file=sys.stderr)
        sys.exit(1)
    images = [src] if src.is_file() else list_images(src)
    if not images:
        print(f"No image files found in {src}", file=sys.stderr)
        sys.exit(1)
    out_dir = args.output_dir.resolve() if args.output_dir else (src / "results" if src.is_dir() else src.parent)
    print(f"Found {len(images)} images in {src}")
    params = {"max_new_tokens": args.max_new_tokens} if args.max_new_tokens else None
    transcribe_folder(images, out_dir, args.ground_truth_dir, params)
    print(f"\nAll done. Results saved to {out_dir}")


if __name__ == "__main__":
    main()
