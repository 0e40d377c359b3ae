#!/usr/bin/env python3
"""Does PARTITIONING the chip beat sharing it?  One batch's tower + prefill (matrix-pipe bound) on a stream masked to `big` CUs beside
another batch's decode (HBM / latency bound) on a stream masked to the other `small` CUs (hipExtStreamCreateWithCUMask), against one
batch at a time on the whole chip and against two lanes sharing every CU (pipeline.LanePipeline "lockstep").

    python tools/bench_partition.py [--small 64,48] [--pages 84] [--batches 3] [--new-tokens 512] [--share-small 0]

Prints one JSON object: where the masked streams' workgroups really ran (per-XCD CU counts), the phases of a batch alone on the whole
chip / on each partition, and the pipelined rate per split.  Tokens must be identical in every arm.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from handwritten_ocr_amd import _lib, engine, gpupre  # noqa: E402


def masked_stream(lib, bits):
    words = (C.c_uint * 8)(*[sum(1 << b for b in range(32) if 32 * w + b in bits) for w in range(8)])
    h = C.c_void_p()
    _lib.check(lib.hwocr_stream_create_cumask(words, 8, C.byref(h)), "hwocr_stream_create_cumask")
    return torch.cuda.ExternalStream(h.value), h


def placement(lib, stream, dev):
    out = torch.zeros(4096, 2, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        _lib.check(lib.hwocr_probe_placement(_lib.ptr(out), 4096, 40000, _lib.stream_handle()), "hwocr_probe_placement")
        stream.synchronize()
    o = out.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    xcc, hw = o[:, 0] & 0xF, o[:, 1]
    cu = (hw >> 8) & 0xFF  # cu [11:8], sh [12], se [15:13]
    per = {}
    for x, c in zip(xcc.tolist(), cu.tolist()):
        per.setdefault(int(x), set()).add(int(c))
    return {"cus_seen": int(sum(len(v) for v in per.values())), "per_xcd": {k: len(v) for k, v in sorted(per.items())}}


class PartitionHooks:
    """tower + prefill on `big`, decode on `small`; the locks keep each stream's batches whole and in order."""

    def __init__(self, lib, big, small, n_big, n_small, big_lock, small_lock):
        self.lib, self.big, self.small, self.n_big, self.n_small = lib, big, small, n_big, n_small
        self.big_lock, self.small_lock = big_lock, small_lock

    def tower_begin(self):
        self.big_lock.acquire()
        torch.cuda.set_stream(self.big)
        self.lib.hwocr_set_cu_budget(self.n_big)

    def prefill_end(self):
        ev = torch.cuda.Event()
        ev.record(self.big)
        self.big_lock.release()
        self.small_lock.acquire()
        torch.cuda.set_stream(self.small)
        self.small.wait_event(ev)
        self.lib.hwocr_set_cu_budget(self.n_small)

    def decode_end(self):
        self.small_lock.release()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=84)
    ap.add_argument("--batches", type=int, default=3, help="batches per lane in each pipelined arm")
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--model", default="qwen2-vl-2b")
    ap.add_argument("--small", default="64", help="comma list: CUs of the decode partition")
    ap.add_argument("--share-small", type=int, default=0, help="1: the decode stream may use EVERY CU (only the tower is confined)")
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--lockstep", type=int, default=1, help="1: also time the lanes sharing every CU, started together")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.hip()
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

    def one(eng, hooks=None):
        return eng.generate(pages, prompts, max_new=args.new_tokens, min_new=args.new_tokens, hooks=hooks)

    for e in engs:
        e.collect_timings = True
        want = one(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        got = one(engs[0])
    torch.cuda.synchronize()
    serial_s = (time.perf_counter() - t0) / 2
    assert got == want
    out = {"pages_per_batch": args.pages, "new_tokens": args.new_tokens,
           "whole_chip_one_batch_at_a_time": {"ms_per_batch": 1e3 * serial_s, "pages_per_s": args.pages / serial_s, "phases_ms": dict(engs[0].timings)},
           "splits": []}
    # two lanes sharing every CU, started together (pipeline.LanePipeline "lockstep"), same batch count
    if args.lockstep:
        res = [None] * args.lanes

        def free(i):
            torch.cuda.set_device(dev)
            s_ = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s_):
                for _ in range(args.batches):
                    res[i] = one(engs[i])
                s_.synchronize()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=free, args=(i,)) for i in range(args.lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["lockstep_lanes"] = {"batches": args.lanes * args.batches, "seconds": dt, "pages_per_s": args.lanes * args.batches * args.pages / dt,
                                 "tokens_identical": bool(all(r == want for r in res))}
        print(json.dumps(out["lockstep_lanes"]), file=sys.stderr, flush=True)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    for n_small in [int(v) for v in args.small.split(",")]:
        n_big = ncu - n_small
        small_bits = set(range(ncu)) if args.share_small else set(range(n_small))
        big_bits = set(range(n_small, ncu))
        big, hb = masked_stream(lib, big_bits)
        small, hs = masked_stream(lib, small_bits)
        rec = {"small_cus": n_small, "big_cus": n_big, "decode_may_use_every_cu": bool(args.share_small),
               "placement_big": placement(lib, big, dev), "placement_small": placement(lib, small, dev)}
        # a batch alone, its two halves on their partitions (nothing else on the chip)
        locks = (threading.Lock(), threading.Lock())
        hooks = PartitionHooks(lib, big, small, n_big, n_small if not args.share_small else 0, *locks)
        cur = torch.cuda.current_stream()
        got = one(engs[0], hooks)
        torch.cuda.synchronize()
        rec["alone_on_partitions_phases_ms"] = dict(engs[0].timings)
        rec["tokens_identical"] = bool(got == want)
        torch.cuda.set_stream(cur)
        lib.hwocr_set_cu_budget(0)
        # pipelined
        results, phases, finished = [None] * args.lanes, [[] for _ in range(args.lanes)], []

        def worker(i):
            torch.cuda.set_device(dev)
            h = PartitionHooks(lib, big, small, n_big, n_small if not args.share_small else 0, *locks)
            for _ in range(args.batches):
                results[i] = one(engs[i], h)
                finished.append(time.perf_counter())
                phases[i].append(dict(engs[i].timings))
            lib.hwocr_set_cu_budget(0)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(args.lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nb = args.lanes * args.batches
        mid = lambda i, k: float(np.median([p[k] for p in phases[i][1:]] or [phases[i][0][k]]))  # noqa: E731
        gaps = np.diff(sorted(finished))
        steady = float(np.median(gaps[1:])) if len(gaps) > 2 else float("nan")
        rec["pipelined"] = {"batches": nb, "seconds": dt, "pages_per_s": nb * args.pages / dt,
                            "batch_completion_gaps_s": [round(float(g), 3) for g in gaps],
                            "steady_state": {"s_per_batch": steady, "pages_per_s": args.pages / steady,
                                             "note": "median gap between consecutive batch completions, first gap dropped"},
                            "speedup_vs_one_at_a_time": (nb * args.pages / dt) / (args.pages / serial_s),
                            "phases_ms": [{k: mid(i, k) for k in ("vision_ms", "prefill_ms", "decode_ms")} for i in range(args.lanes)],
                            "tokens_identical": bool(all(r == want for r in results))}
        out["splits"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        _lib.check(lib.hwocr_stream_destroy(hb), "hwocr_stream_destroy")
        _lib.check(lib.hwocr_stream_destroy(hs), "hwocr_stream_destroy")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
