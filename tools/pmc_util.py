#!/usr/bin/env python3
"""Per-kernel pipe utilisation from a rocprofv3 --pmc counter_collection.csv: MFMA busy and LDS-array active cycles against
the dispatch's GPU cycles.  SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE are summed over all SIMDs / CUs of the chip;
GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md): utilisation = counter / (GRBM_GUI_ACTIVE / 8 x units), units = 1024
SIMDs (MFMA) or 256 CUs (LDS).  Also the effective shader clock of each kernel = GRBM_GUI_ACTIVE / 8 / dispatch duration (the guide's
"DVFS give-back": reads high on dispatches shorter than ~0.3 ms).  usage: pmc_util.py <counter_collection.csv> [kernel substring ...]"""
import collections
import csv
import sys

want = sys.argv[2:] or ["attn_vit80", "gemm_wide256", "gemm_stream", "attn_decode", "attn_prefill"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = next((w for w in want if w in r["Kernel_Name"]), None)
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in agg.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if not cyc:
        continue
    line = f"{k:14s} launches {cnt[k]:5d}  cycles/launch {cyc / cnt[k]:12.0f}  {dur[k] / cnt[k] / 1e3:9.1f} us  clock {cyc / max(1, dur[k]):4.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        line += f"  MFMA busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} %"
    if "SQ_LDS_IDX_ACTIVE" in c:
        line += f"  LDS array active {100 * c['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):5.1f} %"
    if "SQ_LDS_BANK_CONFLICT" in c:
        line += f"  (bank-conflict cycles {100 * c['SQ_LDS_BANK_CONFLICT'] / (cyc * 256):4.1f} %)"
    print(line)
