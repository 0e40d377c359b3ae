#!/bin/bash
# Run ON the MI355X box: issue / wait / co-execution counters of the ViT attention at the bench's page shape (tools/bench_attn_vit80.py:
# 12 pages x 16 heads x 5184 tokens x head_dim 80), for the shipped one-wave-per-SIMD kernel (x) and the 12-wave form it replaced.
# Counter passes carry --kernel-trace only.  usage: tools/pmc_vit_attn.sh <tag>     -> gpurun_out/<tag>_pmc_vit_attn.txt
set -u
TAG=$1
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_vit_attn.txt
for K in x 12; do
  rm -rf /tmp/pv_a /tmp/pv_b
  HWOCR_VIT80_KERNEL=$K rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU \
      --output-format csv -d /tmp/pv_a -- python3 $R/tools/bench_attn_vit80.py > /dev/null 2> $OUT/${TAG}_pmc_vit_attn.err
  HWOCR_VIT80_KERNEL=$K rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM \
      --output-format csv -d /tmp/pv_b -- python3 $R/tools/bench_attn_vit80.py > /dev/null 2>> $OUT/${TAG}_pmc_vit_attn.err
  python3 - "$K" <<'PY' >> $OUT/${TAG}_pmc_vit_attn.txt
import collections, csv, glob, sys
c = collections.defaultdict(float); n = collections.Counter(); dur = collections.Counter()
for d in ("/tmp/pv_a", "/tmp/pv_b"):
    for fn in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if "attn_vit80" not in r["Kernel_Name"]:
                continue
            c[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur[r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
name = "attn_vit80x_kernel (one wave per SIMD)" if sys.argv[1] == "x" else "attn_vit80_kernel<12> (twelve lockstep waves)"
print(name)
for k in sorted(c):
    print(f"    {k:30s} {c[k] / n[k]:16.0f} per launch  ({n[k]} launches, {dur[k] / n[k] / 1e3:8.1f} us each under the profiler)")
w = c["SQ_WAVE_CYCLES"] / max(1, n["SQ_WAVE_CYCLES"])
if w:
    g = lambda k: c[k] / max(1, n[k])
    print("    share of wave cycles (quad-cycles): waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f %%)" % (
        100 * g("SQ_WAIT_ANY") / w, 100 * g("SQ_WAIT_INST_ANY") / w, 100 * g("SQ_ACTIVE_INST_ANY") / w, 100 * g("SQ_ACTIVE_INST_VALU") / w))
    busy = g("SQ_BUSY_CYCLES")
    print("    matrix pipe busy %.0f cycles per launch; vector and matrix instructions executing TOGETHER %.0f (%.1f %% of the matrix-busy cycles)" % (
        g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_VALU_MFMA_COEXEC_CYCLES"), 100 * g("SQ_VALU_MFMA_COEXEC_CYCLES") / max(1.0, g("SQ_VALU_MFMA_BUSY_CYCLES"))))
    cyc = g("GRBM_GUI_ACTIVE") / 8
    if cyc:
        print("    %.0f shader cycles per launch at %.2f GHz; per SIMD: MFMA busy %.1f %%; instructions per launch: VALU %.3g (transcendental %.3g), MFMA %.3g, LDS %.3g" % (
            cyc, cyc / (dur["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"]), 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024), g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU_TRANS"), g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS")))
PY
done
cat $OUT/${TAG}_pmc_vit_attn.txt
rm -rf /tmp/pv_a /tmp/pv_b
