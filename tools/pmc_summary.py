#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM-side traffic.

Corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
coalesced reads (16 B/lane global loads and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-byte streaming stores.
Infinity-Cache hits are included (the counters sit on the L2's fabric side).

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def key(name: str) -> str:
    if re.search(r"gemm_wide256_kernel<\d+, (true|false), true", name):  # the E4M3 instances (<EPI, STAGGER, FP8, ...>)
        return "gemm_wide256_kernel_fp8"
    if "gemm_wide256w4_kernel" in name:  # the four-wave form of the same GEMM: one family for the traffic figure bench.py reads
        return "gemm_wide256_kernel"
    m = re.search(r"(gemm_wide256_kernel|gemm_wide_kernel|gemm_skinny_kernel|attn_vit80_kernel|attn_prefill_kernel|"
                  r"attn_decode_kernel|attn_decode_merge_kernel|\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def load(path: str, counter: str):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[key(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main():
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch (mean over the launches of the profiled command)",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads); includes Infinity-Cache hits",
           "kernels": {}, "wide_gemm_forms": {}}
    # the two forms of the wide GEMM on their own (the family entry above is what bench.py's roofline.traffic reads)
    global key
    family = key
    key = lambda name: ("gemm_wide256w4_kernel (four waves)" if "gemm_wide256w4_kernel" in name else  # noqa: E731
                        "gemm_wide256_kernel (eight waves)" if family(name) == "gemm_wide256_kernel" else "")
    fe2, wr2 = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    key = family
    for k in fe2:
        if k and fe2[k][0]:
            n, f, us = fe2[k]
            wn, w, _ = wr2.get(k, [0, 0.0, 0.0])
            out["wide_gemm_forms"][k] = {"launches": n, "read_bytes": 2.0 * f / n * 1024.0, "write_bytes": (w / wn * 1024.0) if wn else 0.0,
                                         "avg_us_under_pmc": us / n}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n, f, us = fe[k]
        wn, w, _ = wr.get(k, [0, 0.0, 0.0])
        if n == 0:
            continue
        rd = 2.0 * f / n * 1024.0
        wt = (w / wn * 1024.0) if wn else 0.0
        out["kernels"][k] = {"launches": n, "read_bytes": rd, "write_bytes": wt, "traffic_bytes": rd + wt,
                             "avg_us_under_pmc": us / n}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in list(out["kernels"].items())[:12]:
        print(f"{k:28s} n={v['launches']:5d} traffic {v['traffic_bytes'] / 1e6:9.1f} MB/launch  {v['avg_us_under_pmc']:8.1f} us")


if __name__ == "__main__":
    main()
