#!/usr/bin/env python3
"""does running one batch's tower + prefill (matrix-pipe bound) BESIDE another batch's decode (HBM / latency bound)
on two HIP streams beat running them back to back?

Two ReadEngines over the SAME weight tensors (two KV caches, two workspaces, 252 read slots each), each driven by its own host
thread on its own stream; engine B starts half a step after engine A, so that in the steady state A's decode runs beside B's tower
+ prefill and vice versa.  Compared with ONE engine doing the same number of batches back to back (the bench's schedule).
Reports pages/s both ways, the per-phase wall times inside the overlapped run, and the tokens (must be identical).

    python tools/bench_overlap.py [--pages 84] [--batches 4] [--new-tokens 512] [--lanes 2] [--offsets 0,0.55]
(CU-masked streams - a partition of the chip between the two halves - are tools/bench_partition.py)
"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from handwritten_ocr_amd import engine, gpupre  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=84)
    ap.add_argument("--batches", type=int, default=4, help="batches per engine in the overlapped run (the serial run does 2x)")
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--model", default="qwen2-vl-2b")
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--offsets", default="0.55", help="comma list: start of lane i delayed by i x offset x (one serial step)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = engine.preset(args.model)
    strategies = bench.strategies_for(3)
    hw = bench.target_hw(cfg, 1024)
    n_img = (hw[0] // cfg.patch_size) * (hw[1] // cfg.patch_size) // cfg.merge ** 2
    n_reads = args.pages * 3
    sd = engine.random_state_dict(cfg, seed=0, device=dev)
    eng0 = engine.ReadEngine(cfg, sd, max_reads=n_reads, ctx=2048, device=str(dev), vit_batch=12, prefill_batch=16)
    engs = [eng0] + [eng0.lane() for _ in range(args.lanes - 1)]
    del sd
    sp = gpupre.StrategyPages(dev)
    raws = [torch.from_numpy(r).to(dev) for r in bench.raw_pages(args.pages, 0, 1024)]
    pages = [im for raw in raws for im in sp.pages(raw, strategies, hw)]
    prompts = [bench.synthetic_prompt(cfg, n_img)] * n_reads
    torch.cuda.synchronize()

    def one(eng):
        return eng.generate(pages, prompts, max_new=args.new_tokens, min_new=args.new_tokens)

    for e in engs:  # warm-up: graph capture, lazy kernel attributes
        e.collect_timings = True
        want = one(e)
    torch.cuda.synchronize()

    # ---- serial: one engine, 2 x batches back to back
    t0 = time.perf_counter()
    for _ in range(2 * args.batches):
        got = one(engs[0])
    torch.cuda.synchronize()
    serial_s = time.perf_counter() - t0
    assert got == want
    serial_phases = dict(engs[0].timings)

    # ---- overlapped: one lane per host thread and stream, lane i starting i x offset steps late, NO ordering between the lanes
    import json

    step_s = serial_s / (2 * args.batches)
    out = {"pages_per_batch": args.pages, "new_tokens": args.new_tokens, "lanes": args.lanes,
           "serial": {"batches": 2 * args.batches, "seconds": serial_s, "pages_per_s": 2 * args.batches * args.pages / serial_s, "phases_ms": serial_phases},
           "free_running": []}
    for off in [float(v) for v in args.offsets.split(",")]:
        results, phases = [None] * args.lanes, [[] for _ in range(args.lanes)]

        def worker(i):
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                if i:
                    time.sleep(step_s * off * i)
                for _ in range(args.batches):
                    results[i] = one(engs[i])
                    phases[i].append(dict(engs[i].timings))
                s.synchronize()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(args.lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        overlap_s = time.perf_counter() - t0
        nb = args.lanes * args.batches
        mid = lambda i, k: float(np.median([p[k] for p in phases[i][1:]] or [phases[i][0][k]]))  # noqa: E731
        out["free_running"].append({"offset": off, "batches": nb, "seconds": overlap_s, "pages_per_s": nb * args.pages / overlap_s,
                                    "speedup_vs_serial": (nb * args.pages / overlap_s) / (2 * args.batches * args.pages / serial_s),
                                    "phases_ms": [{k: mid(i, k) for k in ("vision_ms", "prefill_ms", "decode_ms")} for i in range(args.lanes)],
                                    "tokens_identical_to_serial": bool(all(r == want for r in results))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
