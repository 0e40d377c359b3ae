#!/usr/bin/env python3
"""BASELINE config 3's path on the hardware at hand: a folder of synthetic 1024x1024 handwritten pages through the batch-folder
driver (`handwritten_ocr_amd.batch.transcribe_folder`) — PNG decode on host threads, strategy preprocessing + resize on the
device, continuous batching through the engine's decode slots, detokenise, `initial_ocr` replay (compare / merge), the four
output files per page.  Random-init Qwen2-VL-2B shape, fixed 512 tokens per read (min_new == max_new) like bench.py.

    python tools/bench_folder.py [pages=256]            one process, one GPU
    HWOCR_ALLOW_SHARED_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/bench_folder.py 64
                                                        (rehearsal of the sharded form on a one-GPU box)"""
import contextlib
import io
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("HWOCR_MODEL", "qwen2-vl-2b")
os.environ.setdefault("HWOCR_ALLOW_RANDOM_INIT", "1")
os.environ.setdefault("HWOCR_GPU_PREPROCESS", "1")
os.environ.setdefault("HWOCR_CTX", "2048")

from PIL import Image  # noqa: E402

from handwritten_ocr_amd import batch, shard, synth, tools  # noqa: E402

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rank, _, world = shard.init_from_env()
root = Path(os.environ.get("HWOCR_BENCH_DIR", tempfile.gettempdir())) / "hwocr_folder_bench"
src, out = root / "pages", root / "results"
if rank == 0:
    src.mkdir(parents=True, exist_ok=True)
    for i in range(n_pages):
        f = src / f"page{i:04d}.png"
        if not f.exists():
            Image.fromarray(synth.make_page(i, 1024, 1024), "RGB").save(f, compress_level=1)
if world > 1:
    import torch.distributed as dist

    dist.barrier()
images = batch.list_images(src)[:n_pages]
params = {"max_new_tokens": 512, "min_new_tokens": 512}
with contextlib.redirect_stdout(io.StringIO()):
    tools._load_ocr_model()                                   # weights resident before the clock, as in the reference's 2nd page on
    batch.transcribe_folder(images[: 4 * world], out, params=params, quiet=True)   # warm-up: graphs, LDS attributes
stats = {}
model, _ = tools._load_ocr_model()
if os.environ.get("HWOCR_FOLDER_TRACE"):
    model.collect_timings = True
    n_lanes, _ = tools.plan_lanes(3 * n_pages // world, model.max_reads, int(os.environ.get("HWOCR_LANES", "2")))
    for e in (tools._lanes(n_lanes).engines if n_lanes > 1 else [model]):
        e.collect_timings = True
t0 = time.perf_counter()
outs = batch.transcribe_folder(images, out, params=params, quiet=True, stats=stats)
dt = time.perf_counter() - t0
if rank == 0:
    assert len(outs) == len(images) and all(o.exists() for o in outs)
    print(f"{len(images)} pages x 3 strategy reads x 512 tokens through transcribe_folder on {world} rank(s): {dt:.2f} s = "
          f"{len(images) / dt:.2f} pages/s (4 files per page written; preprocessing on the device: {os.environ['HWOCR_GPU_PREPROCESS']}); "
          f"rank 0 host work {1e3 * stats['rank0_host_s'] / len(images):.2f} ms per page in {stats['gather_rounds']} gather rounds, "
          f"{1e3 * stats['writer_tail_s']:.0f} ms of it left after the last read stopped")
    print(f"  read_pages: host preparation {batch.LAST_READ_TIMINGS.get('prepare_s', 0):.2f} s, engine pass {batch.LAST_READ_TIMINGS.get('engine_s', 0):.2f} s")
    if os.environ.get("HWOCR_FOLDER_TRACE"):
        engines = tools._lanes(n_lanes).engines[:n_lanes] if n_lanes > 1 else [model]
        for li, e in enumerate(engines):
            tr = getattr(e, "stream_trace", [])
            adm = [(n, round(ms)) for n, ms in tr if n]
            dec = [ms for n, ms in tr if not n]
            print(f"  lane {li}: {len(tr)} trips; admitting trips (reads, ms): {adm}; {len(dec)} decode-only trips, mean {sum(dec) / max(1, len(dec)):.1f} ms")
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
