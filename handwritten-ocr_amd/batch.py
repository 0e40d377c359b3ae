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

Multi-GPU: one process per GPU (`torchrun`, or any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE), the
sorted page list dealt round-robin (`shard.shard`), all reads of a page on one GPU; every rank runs only the engine pass over
its pages, the generated token streams are gathered to rank 0 over RCCL (`shard.gather_token_streams`, a few KB per read), and
rank 0 alone detokenises, replays the nodes, runs the agents and writes every page's files — the reference's loop
(transcribe.py:185-210) with its engine work spread over the node.

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


def read_pages(image_paths: list, params: dict | None = None, cfg=config, workers: int = 8,
               speculate_reocr: bool = False, page_numbers: list | None = None) -> tuple[list, list]:
    """ONE batched engine pass over every strategy read of `image_paths`: (strategies, streams) with streams[p][k] = the
    generated token ids of page p under strategies[k].  `speculate_reocr`: also read the strategies a later `reocr` node
    would use."""
    strategies = _speculative_strategies(list(cfg.PREPROCESSING_STRATEGIES), every=speculate_reocr)
    if not image_paths:
        return strategies, []

    # HWOCR_GPU_PREPROCESS=1: the strategy chains and the processor's resize run on the device, bit-identical
    # to the host path (gpupre.py); a page is decoded and uploaded once for all its reads.  Pages that are not plain RGB,
    # chains the device path does not cover (OpenCV present, a transform after binarize) and pages whose file format
    # re-quantises on save (below) keep the host path.
    gpu_pages = None
    if os.environ.get("HWOCR_GPU_PREPROCESS", "0") not in ("", "0"):
        from . import gpupre

        if all(gpupre.supported(s) for s in strategies):
            model, processor = tools._load_ocr_model()
            gpu_pages = (gpupre.StrategyPages(model.dev), processor)

    def prepare(path):
        img = Image.open(path)
        img.load()
        suffix = Path(path).suffix.lower()
        lossy = suffix in preprocess.LOSSY_SUFFIXES
        if gpu_pages is not None and img.mode == "RGB" and not lossy:
            return np.asarray(img)
        # The serial path hands every transformed page to the model through a temp file with the INPUT's suffix
        # (tools.py:668-672): for .jpg / .jpeg / .webp that re-encode changes the pixels, so it is reproduced here (in
        # memory); "original" reads the input file itself (tools.py:651-652) and is not re-encoded.
        out = []
        for s in strategies:
            pre = preprocess.apply_strategy(img, s, quiet=True)
            if lossy and preprocess.steps_of(s) not in (["original"], []):
                pre = preprocess.through_tempfile_codec(pre, suffix)
            out.append(pre)
        return out

    with ThreadPoolExecutor(max_workers=workers) as pool:
        prepared = list(pool.map(prepare, image_paths))
    if gpu_pages is not None:
        sp, processor = gpu_pages
        prepared = [sp.pages(p, strategies, processor.target_hw(p.shape[0], p.shape[1])) if isinstance(p, np.ndarray) else p
                    for p in prepared]
    flat = [im for page in prepared for im in page]
    k = len(strategies)
    if page_numbers is not None:  # number the reads by page in the WHOLE job (a rank holds a share): keys the sampling RNG
        params = dict(params or {}, read_ids=[int(n) * k + j for n in page_numbers for j in range(k)])
    toks = tools.run_ocr_batch_tokens(flat, params)
    return strategies, [toks[p * k: (p + 1) * k] for p in range(len(image_paths))]


def replay_initial_ocr(image_paths: list, strategies: list, texts: list, cfg=config) -> list[dict]:
    """States after `initial_ocr` for every page: the reference's node code run per page with `preprocess_image` /
    `run_ocr` answering from the finished batch (texts[p][k] = page p under strategies[k])."""
    states = []
    for path, page_texts in zip(image_paths, texts):
        fake_preprocess, fake_run_ocr = _replay_readers(strategies, page_texts)
        state = new_state(str(path), cfg)
        saved = (nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model)
        nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model = fake_preprocess, fake_run_ocr, (lambda: None)
        try:
            state.update(nodes.node_initial_ocr(state))
        finally:
            nodes.preprocess_image, nodes.run_ocr, nodes.unload_ocr_model = saved
        states.append(state)
    return states


def initial_ocr_batched(image_paths: list, params: dict | None = None, cfg=config, workers: int = 8,
                        speculate_reocr: bool = False, reads_out: list | None = None) -> list[dict]:
    """States after `initial_ocr` for every page, computed with ONE batched engine pass over all reads (single process).
    `reads_out` (a list) receives, per page, (strategies, texts) of everything that was read."""
    strategies, streams = read_pages(image_paths, params, cfg, workers, speculate_reocr)
    flat = tools.decode_tokens([t for page in streams for t in page])
    k = len(strategies)
    texts = [flat[p * k: (p + 1) * k] for p in range(len(image_paths))]
    if reads_out is not None:
        reads_out.extend((strategies, t) for t in texts)
    return replay_initial_ocr(image_paths, strategies, texts, cfg)


def gather_reads(streams: list, n_pages_total: int, k: int) -> list | None:
    """Every rank's streams[p][k] (pages dealt round-robin, k reads per page) -> on rank 0 the streams of ALL pages in the global page order;
    None elsewhere.  One padded fixed-shape gather (shard.gather_token_streams): RCCL on the GPUs, gloo in the CPU tests."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return streams
    world = dist.get_world_size()
    dev = torch.device(f"cuda:{torch.cuda.current_device()}") if dist.get_backend() == "nccl" else torch.device("cpu")
    flat = [t for page in streams for t in page]
    width = max([len(t) for t in flat] + [1])
    toks = torch.zeros(len(flat), width, dtype=torch.int32)
    for i, t in enumerate(flat):
        toks[i, : len(t)] = torch.tensor(t, dtype=torch.int32)
    counts = torch.tensor([len(t) for t in flat], dtype=torch.int32)
    got = shard.gather_token_streams(toks.to(dev), counts.to(dev), dst=0)
    if got is None:
        return None
    per_rank = []
    for t, c in got:
        t, c = t.cpu().tolist(), c.cpu().tolist()
        per_rank.append([row[:n] for row, n in zip(t, c)])
    out = []
    for r, j in shard.owner_index(n_pages_total, world):
        out.append(per_rank[r][j * k: (j + 1) * k])
    return out


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
    """The batch folder: every rank reads its share of `images` in one batched engine pass; rank 0 gathers the token
    streams and does the rest for all pages — detokenise, `initial_ocr` replay, (with agents) the critic / editor / re-OCR
    loop, the four output files per page.  Returns the transcription paths on rank 0, [] elsewhere."""
    rank, _, world = shard.init_from_env()
    images = [Path(p) for p in images]
    mine = shard.shard(images, rank, world)
    sink = io.StringIO() if quiet else None
    err = None
    try:
        with contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext():
            strategies, streams = read_pages([str(p) for p in mine], params, speculate_reocr=bool(agents),
                                             page_numbers=shard.shard(list(range(len(images))), rank, world))
    except Exception as e:  # noqa: BLE001  (reported to every rank below, then re-raised here)
        err = e
    if world > 1:
        # a rank whose reads raised (unreadable image, HwocrError) must not leave the others waiting in the gather: everyone learns
        # who failed first, and every rank leaves with an error
        import torch
        import torch.distributed as dist

        dev = torch.device(f"cuda:{torch.cuda.current_device()}") if dist.get_backend() == "nccl" else None
        bad = shard.failed_ranks(err is None, dev)
        if bad and err is None:
            raise RuntimeError(f"rank(s) {bad} failed while reading their pages; rank {rank} stops with them")
    if err is not None:
        raise err
    streams = gather_reads(streams, len(images), len(strategies))
    if streams is None:  # not rank 0: its reads are on their way to rank 0
        return []
    flat = tools.decode_tokens([t for page in streams for t in page])
    k = len(strategies)
    texts = [flat[p * k: (p + 1) * k] for p in range(len(images))]
    with contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext():
        states = replay_initial_ocr([str(p) for p in images], strategies, texts)
    outs = []
    for state, page_texts in zip(states, texts):
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
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        print(f"Found {len(images)} images in {src}")
    params = {"max_new_tokens": args.max_new_tokens} if args.max_new_tokens else None
    transcribe_folder(images, out_dir, args.ground_truth_dir, params)
    if rank == 0:
        print(f"\nAll done. Results saved to {out_dir}")
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():  # ranks leave together (rank 0 is still writing while the others are done)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
