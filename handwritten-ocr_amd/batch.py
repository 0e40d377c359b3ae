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

Host work beside the reads (round 4): the engine reports every read as it stops (`generate_stream(on_done=)`), a page is complete
when its reads are in, and complete pages leave in ROUNDS of `GATHER_PAGES` local pages: ONE gather thread per rank issues every
collective of the job, in round order (all ranks make the same number of rounds, an empty one where a rank has fewer pages), while
the lanes go on reading; on rank 0 a writer thread takes the gathered rounds and detokenises / replays / runs the agents / writes the
four files in page order.  So rank 0's file I/O overlaps its own reads, and the other ranks never wait for it: they hand over a
round and read on.

CLI:  python -m handwritten_ocr_amd.batch <folder-or-image> [--output-dir D] [--ground-truth-dir G] [--max-new-tokens N]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import queue
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from PIL import Image

from . import preprocess, shard, tools
from .compat import config, nodes
from .compat.state import new_state

IMAGE_EXTENSIONS = {".png", ".jpg", ".jpeg", ".bmp", ".tiff", ".tif", ".webp"}
LAST_READ_TIMINGS: dict = {}   # read_pages' last call on this process: host preparation / engine pass seconds (tools/bench_folder.py)
GATHER_PAGES = 16   # local pages per gather round: a round's three small collectives cost well under a millisecond over xGMI, a page's
                    # reads ~8 KB each; rank 0's writer starts on a round as soon as the slowest rank has finished those pages


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


class _LazyDeviceReads:
    """The strategy reads of a list of page files as a sequence whose items are MADE WHEN A LANE ADMITS THEM (engine.generate_stream
    indexes it at admission time): the page files are decoded by a pool of host threads in the background - in page order, so the
    first lanes' first pages are ready first - and a page's strategy images are computed on the device (gpupre.StrategyPages: bit
    for bit the host path's pixels) by the lane thread that first needs one of them, on that lane's stream.  Nothing of a 256-page
    folder's half second of PNG decoding then stands in front of the first tower launch.  What the driver needs BEFORE a page is
    decoded - its size, hence its image-token count and prompt - comes from the file header (`image_tokens`).
    Pages that cannot take the device path (not plain RGB; a format that re-quantises on save, whose temp-file re-encode the serial
    path implies: tools.py:668-672) are prepared on the host by the same pool, at the tower's resolution."""

    def __init__(self, paths: list, strategies: list, processor, device, workers: int, prompt: str):
        from . import gpupre

        self.paths, self.strategies, self.k = list(paths), strategies, len(strategies)
        self.processor, self.device, self.prompt = processor, device, prompt
        self._sp = threading.local()          # a lane's own scratch (StrategyPages is not re-entrant)
        self._made: dict = {}                 # (thread, page) -> its k images (a lane reads what it made, on its own stream)
        self._gpupre = gpupre
        self.hw, self.on_device = [], []
        for p in self.paths:
            with Image.open(p) as im:         # header only
                w, h = im.size
                self.on_device.append(im.mode == "RGB" and Path(p).suffix.lower() not in preprocess.LOSSY_SUFFIXES)
            self.hw.append(processor.target_hw(h, w))
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._raw = [self._pool.submit(self._decode, i) for i in range(len(self.paths))]

    def _decode(self, i: int):
        img = Image.open(self.paths[i])
        img.load()
        if self.on_device[i]:
            return np.asarray(img)
        suffix = Path(self.paths[i]).suffix.lower()
        out = []
        for s in self.strategies:
            pre = preprocess.apply_strategy(img, s, quiet=True)
            if suffix in preprocess.LOSSY_SUFFIXES and preprocess.steps_of(s) not in (["original"], []):
                pre = preprocess.through_tempfile_codec(pre, suffix)
            out.append(self.processor.prepare(pre, self.prompt)[0])
        return out

    def __len__(self) -> int:
        return len(self.paths) * self.k

    def shape_of(self, r: int) -> tuple[int, int]:
        return self.hw[r // self.k]

    def image_tokens(self, r: int) -> int:
        c = self.processor.cfg
        h, w = self.hw[r // self.k]
        return (h // c.patch_size) * (w // c.patch_size) // c.merge ** 2

    def __getitem__(self, r: int):
        p, j = divmod(r, self.k)
        key = (threading.get_ident(), p)
        made = self._made.get(key)
        if made is None:
            raw = self._raw[p].result()
            if self.on_device[p]:
                sp = getattr(self._sp, "pages", None)
                if sp is None:
                    sp = self._sp.pages = self._gpupre.StrategyPages(self.device)
                made = sp.pages(raw, self.strategies, self.hw[p])
            else:
                made = raw
            self._made[key] = made
        return made[j]

    def close(self) -> None:
        self._pool.shutdown(wait=False, cancel_futures=True)


def _host_workers() -> int:
    """Threads for decoding page files: the cores this process may use, at most 16 (a GPU's share of the node's host cores)."""
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def read_pages(image_paths: list, params: dict | None = None, cfg=config, workers: int = 0,
               speculate_reocr: bool = False, page_numbers: list | None = None, on_read=None) -> tuple[list, list]:
    """ONE batched engine pass over every strategy read of `image_paths`: (strategies, streams) with streams[p][k] = the
    generated token ids of page p under strategies[k].  `speculate_reocr`: also read the strategies a later `reocr` node
    would use.  `on_read(p * len(strategies) + k, tokens)`: called as soon as that read has stopped (tools.run_ocr_batch_tokens)."""
    strategies = _speculative_strategies(list(cfg.PREPROCESSING_STRATEGIES), every=speculate_reocr)
    if not image_paths:
        return strategies, []

    # HWOCR_GPU_PREPROCESS=1: the strategy chains and the processor's resize run on the device, bit-identical
    # to the host path (gpupre.py); a page is decoded and uploaded once for all its reads, WHILE the engine is already reading the
    # pages before it (_LazyDeviceReads).  Pages that are not plain RGB and pages whose file format re-quantises on save keep the
    # host arithmetic (inside the same lazy sequence); chains the device path does not cover (OpenCV present, a transform after
    # binarize) take the host path below for the whole folder.
    k = len(strategies)
    if page_numbers is not None:  # number the reads by page in the WHOLE job (a rank holds a share): keys the sampling RNG
        params = dict(params or {}, read_ids=[int(n) * k + j for n in page_numbers for j in range(k)])
    if os.environ.get("HWOCR_GPU_PREPROCESS", "0") not in ("", "0"):
        from . import gpupre

        if all(gpupre.supported(s) for s in strategies):
            model, processor = tools._load_ocr_model()
            t_prep = time.perf_counter()
            lazy = _LazyDeviceReads(image_paths, strategies, processor, model.dev, workers or _host_workers(),
                                    (params or {}).get("prompt", cfg.OCR_PROMPT))
            t_eng = time.perf_counter()
            try:
                toks = tools.run_ocr_batch_tokens(lazy, params) if on_read is None else tools.run_ocr_batch_tokens(lazy, params, on_done=on_read)
            finally:
                lazy.close()
            LAST_READ_TIMINGS.update(prepare_s=t_eng - t_prep, engine_s=time.perf_counter() - t_eng, pages=len(image_paths))
            return strategies, [toks[p * k: (p + 1) * k] for p in range(len(image_paths))]

    def prepare(path):
        img = Image.open(path)
        img.load()
        suffix = Path(path).suffix.lower()
        lossy = suffix in preprocess.LOSSY_SUFFIXES
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

    t_prep = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers or _host_workers()) as pool:
        prepared = list(pool.map(prepare, image_paths))
    flat = [im for page in prepared for im in page]
    t_eng = time.perf_counter()
    toks = tools.run_ocr_batch_tokens(flat, params) if on_read is None else tools.run_ocr_batch_tokens(flat, params, on_done=on_read)
    LAST_READ_TIMINGS.update(prepare_s=t_eng - t_prep, engine_s=time.perf_counter() - t_eng, pages=len(image_paths))
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


def initial_ocr_batched(image_paths: list, params: dict | None = None, cfg=config, workers: int = 0,
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


class _PageCollector:
    """This rank's reads arrive in any order (the engine's on_done callback, from one or two lane threads); a page is complete when
    its k reads are in; `wait_pages(hi)` returns once every local page below hi is complete (True) or the reads failed (False)."""

    def __init__(self, n_pages: int, k: int):
        self.cv = threading.Condition()
        self.k = k
        self.streams = [[None] * k for _ in range(n_pages)]
        self.left = [k] * n_pages
        self.complete_upto = 0          # every page below this index is complete
        self.error: BaseException | None = None

    def read_done(self, r: int, seq: list) -> None:
        p, j = divmod(r, self.k)
        with self.cv:
            if self.streams[p][j] is None:
                self.left[p] -= 1
            self.streams[p][j] = seq
            while self.complete_upto < len(self.left) and self.left[self.complete_upto] == 0:
                self.complete_upto += 1
            self.cv.notify_all()

    def fail(self, e: BaseException) -> None:
        with self.cv:
            self.error = e
            self.cv.notify_all()

    def wait_pages(self, hi: int) -> bool:
        with self.cv:
            self.cv.wait_for(lambda: self.complete_upto >= hi or self.error is not None)
            return self.error is None


def _gather_round(pages: list, k: int, dev) -> list | None:
    """One round: this rank's complete pages (each a list of k token streams; possibly none) -> on rank 0 a list over ranks of
    their pages of the round, None elsewhere.  One padded fixed-shape gather (shard.gather_token_streams)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [pages]
    flat = [t for page in pages for t in page]
    width = max([len(t) for t in flat] + [1])
    toks = torch.zeros(len(flat), width, dtype=torch.int32)
    for i, t in enumerate(flat):
        toks[i, : len(t)] = torch.tensor(t, dtype=torch.int32)
    counts = torch.tensor([len(t) for t in flat], dtype=torch.int32)
    got = shard.gather_token_streams(toks.to(dev), counts.to(dev), dst=0)
    if got is None:
        return None
    out = []
    for t, c in got:
        rows = [row[:n] for row, n in zip(t.cpu().tolist(), c.cpu().tolist())]
        out.append([rows[j * k: (j + 1) * k] for j in range(len(rows) // k)])
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


def _finish_page(state: dict, strategies: list, page_texts: list, agents: dict | None, output_dir: Path,
                 ground_truth_dir: Path | None) -> Path:
    """After `initial_ocr`: (with agents) the critic / editor / re-OCR loop, then the four files."""
    if agents:
        nodes.run_critic, nodes.run_editor = agents["critic"], agents["editor"]
        nodes.run_arbitrator = agents.get("arbitrator")
        # re-OCR rounds read the strategies that were prefetched in the batched pass; anything else goes to the engine
        saved = (nodes.preprocess_image, nodes.run_ocr)
        nodes.preprocess_image, nodes.run_ocr = _replay_readers(strategies, page_texts, fallback=saved)
        try:
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
    return write_outputs(state, output_dir, gt)


def transcribe_folder(images: list, output_dir: Path, ground_truth_dir: Path | None = None, params: dict | None = None,
                      agents: dict | None = None, quiet: bool = False, stats: dict | None = None) -> list[Path]:
    """The batch folder: every rank reads its share of `images` in one batched engine pass; complete pages leave in rounds
    (module docstring): the rank's gather thread hands them to rank 0, whose writer thread does the rest for all pages in page order
    — detokenise, `initial_ocr` replay, (with agents) the critic / editor / re-OCR loop, the four output files per page — WHILE the
    lanes read on.  Returns the transcription paths on rank 0, [] elsewhere.  `stats` (a dict) receives rank 0's host seconds
    (`rank0_host_s`: detokenise + replay + agents + files, the work that must keep up with every GPU of the node), `pages`,
    `gather_rounds` and the seconds the writer still needed after the last read had stopped (`writer_tail_s`)."""
    rank, _, world = shard.init_from_env()
    images = [Path(p) for p in images]
    n_total = len(images)
    mine = shard.shard(images, rank, world)
    strategies = _speculative_strategies(list(config.PREPROCESSING_STRATEGIES), every=bool(agents))
    k = len(strategies)
    local_max = (n_total + world - 1) // world
    n_rounds = (local_max + GATHER_PAGES - 1) // GATHER_PAGES        # the same on every rank: the collectives of the job
    col = _PageCollector(len(mine), k)
    rounds: queue.Queue = queue.Queue()                              # rank 0: gathered rounds for the writer
    outs: list = [None] * n_total
    shared = {"bad": [], "writer_error": None, "gather_error": None, "host_s": 0.0}
    dev = None
    if world > 1:
        import torch
        import torch.distributed as dist

        dev = torch.device(f"cuda:{torch.cuda.current_device()}") if dist.get_backend() == "nccl" else torch.device("cpu")

    def gather_loop() -> None:
        # the ONLY thread of this rank that issues collectives, one round after the other (every backend wants one order per rank)
        try:
            if dev is not None and dev.type == "cuda":
                import torch

                torch.cuda.set_device(dev)
            for j in range(n_rounds):
                lo, hi = min(j * GATHER_PAGES, len(mine)), min((j + 1) * GATHER_PAGES, len(mine))
                ok = col.wait_pages(hi)
                if world > 1:
                    # a rank whose reads raised must not leave the others waiting in the gather: everyone learns who failed first
                    bad = shard.failed_ranks(ok, dev)
                    if bad:
                        shared["bad"] = bad
                        return
                elif not ok:
                    return
                got = _gather_round(col.streams[lo:hi], k, dev)
                if rank == 0:
                    rounds.put((j, got))
        except BaseException as e:  # noqa: BLE001  (re-raised on the main thread)
            shared["gather_error"] = e
        finally:
            rounds.put(None)

    def writer_loop() -> None:
        try:
            while True:
                item = rounds.get()
                if item is None:
                    return
                j, per_rank = item
                t0 = time.perf_counter()
                for loc in range(j * GATHER_PAGES, (j + 1) * GATHER_PAGES):
                    for r in range(world):
                        i = loc * world + r                      # pages are dealt round-robin: global page i = local page loc of rank r
                        if i >= n_total:
                            continue
                        page_streams = per_rank[r][loc - j * GATHER_PAGES]
                        page_texts = tools.decode_tokens(page_streams)
                        state = replay_initial_ocr([str(images[i])], strategies, [page_texts])[0]
                        outs[i] = _finish_page(state, strategies, page_texts, agents, output_dir, ground_truth_dir)
                shared["host_s"] += time.perf_counter() - t0
        except BaseException as e:  # noqa: BLE001
            shared["writer_error"] = e
            while rounds.get() is not None:                          # keep draining so that the gather thread never blocks on us
                pass

    err = None
    with contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext():   # (sys.stdout is process-wide: all threads)
        gth = threading.Thread(target=gather_loop, name="hwocr-gather")
        wth = threading.Thread(target=writer_loop, name="hwocr-writer") if rank == 0 else None
        gth.start()
        if wth is not None:
            wth.start()
        try:
            got_strategies, _ = read_pages([str(p) for p in mine], params, speculate_reocr=bool(agents),
                                           page_numbers=shard.shard(list(range(n_total)), rank, world), on_read=col.read_done)
            if [nodes._strategy_label(s) for s in got_strategies] != [nodes._strategy_label(s) for s in strategies]:
                raise RuntimeError("read_pages read other strategies than the driver planned for")
            with col.cv:                                             # (an engine that returns without calling back: complete now)
                missing = col.complete_upto < len(mine)
            if missing:
                raise RuntimeError("the engine returned before reporting every read (on_done)")
        except BaseException as e:  # noqa: BLE001  (reported to every rank by the gather thread, then re-raised here)
            err = e
            col.fail(e)
        t_reads = time.perf_counter()
        gth.join()
        if wth is not None:
            wth.join()
    if err is not None:
        raise err
    if shared["bad"]:
        raise RuntimeError(f"rank(s) {shared['bad']} failed while reading their pages; rank {rank} stops with them")
    for key in ("gather_error", "writer_error"):
        if shared[key] is not None:
            raise shared[key]
    if stats is not None:
        stats.update(rank0_host_s=shared["host_s"], pages=n_total, gather_rounds=n_rounds, writer_tail_s=time.perf_counter() - t_reads)
    return [o for o in outs if o is not None] if rank == 0 else []


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
