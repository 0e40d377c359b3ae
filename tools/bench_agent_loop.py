#!/usr/bin/env python3
"""BASELINE config 5 on the hardware at hand: the full critic <-> editor <-> reocr loop over a folder of 64 synthetic 1024x1024
pages, accept threshold 90, through `batch.transcribe_folder(agents=...)` on ONE MI355X (the 8-GPU form is the same call under
torchrun; no multi-GPU node is available to the builder).  The LLM agents are out of scope (they talk to an Ollama server): SCRIPTED
stand-ins with the reference's interfaces drive the control flow — the critic asks for one `reocr` per page (confidence 40,
"needs_reocr"), the arbitrator merges, the critic then accepts at 95 >= 90 — so every page walks initial_ocr -> critic -> reocr ->
arbitrate -> critic -> accept (ocr_agent/graph.py:22-43, nodes.py:239-302).  The engine is entered ONCE: with agents attached every
distinct preprocessing strategy of every page is read in the batched pass (5 per page: 320 reads) and the `reocr` node is answered
from it; the reference reloads the checkpoint and runs one serial read per `reocr`.
Random-init Qwen2-VL-2B shape, 512 tokens per read (min_new == max_new), like bench.py.

    python tools/bench_agent_loop.py [pages=64]"""
import contextlib
import io
import json
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
from handwritten_ocr_amd.compat import config  # noqa: E402

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
config.ACCEPT_THRESHOLD = 90                                   # BASELINE config 5
rank, _, world = shard.init_from_env()
root = Path(os.environ.get("HWOCR_BENCH_DIR", tempfile.gettempdir())) / "hwocr_loop_bench"
src, out = root / "pages", root / "results"
if rank == 0:
    src.mkdir(parents=True, exist_ok=True)
    for i in range(n_pages):
        f = src / f"page{i:04d}.png"
        if not f.exists():
            Image.fromarray(synth.make_page(2000 + i, 1024, 1024), "RGB").save(f, compress_level=1)
if world > 1:
    import torch.distributed as dist

    dist.barrier()
images = batch.list_images(src)[:n_pages]


class Arbitrated:
    def __init__(self, versions):
        self.final_text, self.confidence = versions[-1]["text"] + " [arbitrated]", 70
        self.decisions, self.uncertain_segments = [], []

    def model_dump(self):
        return {"final_text": self.final_text, "confidence": self.confidence, "decisions": [], "uncertain_segments": []}


def critic(text, previous_critique=None):
    if text.endswith("[arbitrated]"):   # second visit: the re-read has been arbitrated
        return {"overall_confidence": 95, "verdict": "needs_edit", "issues": []}   # accepted by the threshold (95 >= 90), not by the verdict
    return {"overall_confidence": 40, "verdict": "needs_reocr", "issues": []}


agents = {"critic": critic, "editor": lambda t, c: {"corrected_text": t, "changes": []}, "arbitrator": Arbitrated}
params = {"max_new_tokens": 512, "min_new_tokens": 512}
strategies = batch._speculative_strategies(list(config.PREPROCESSING_STRATEGIES), every=True)
with contextlib.redirect_stdout(io.StringIO()):
    tools._load_ocr_model()
    batch.transcribe_folder(images[: 4 * world], out, params=params, agents=agents, quiet=True)   # warm-up
stats = {}
t0 = time.perf_counter()
outs = batch.transcribe_folder(images, out, params=params, agents=agents, quiet=True, stats=stats)
dt = time.perf_counter() - t0
if rank == 0:
    ev = json.loads((out / f"{images[0].stem}_trace.json").read_text())
    res = json.loads((out / f"{images[0].stem}_eval.json").read_text())
    walk = [e["agent"] + ":" + e["action"] for e in ev]
    print(f"{len(images)} pages x {len(strategies)} distinct strategy reads x 512 tokens, full loop with scripted agents (accept >= {config.ACCEPT_THRESHOLD}) "
          f"on {world} rank(s): {dt:.2f} s = {len(images) / dt:.2f} pages/s; status {res['pipeline_status']!r} after {res['iterations']} iteration(s), "
          f"confidence {res['final_confidence']}; rank 0 host work {1e3 * stats['rank0_host_s'] / len(images):.2f} ms per page")
    print("  one page's walk:", " -> ".join(walk))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
