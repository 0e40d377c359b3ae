#!/bin/bash
# Run ON the MI355X box: SQ wait / issue counters and L1 (TCP) counters of the decode-step GEMMs (gemm_stream_kernel at 252 reads, the
# five GEMMs exactly as hwocr_decode_step issues them, weights cold: tools/bench_decode_plan.py).  Counter passes carry --kernel-trace
# only.  usage: tools/pmc_decode.sh <tag>      -> gpurun_out/<tag>_pmc_decode.txt
set -u
TAG=$1
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1
rm -rf /tmp/pd_sq /tmp/pd_tcp /tmp/pd_tcp2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/pd_sq -- python3 $R/tools/bench_decode_plan.py qwen2-vl-2b 252 > /dev/null 2> $OUT/${TAG}_pmc_decode.err
echo "sq pass done"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d /tmp/pd_tcp -- python3 $R/tools/bench_decode_plan.py qwen2-vl-2b 252 > /dev/null 2>> $OUT/${TAG}_pmc_decode.err
echo "tcp pass done"
rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d /tmp/pd_tcp2 -- python3 $R/tools/bench_decode_plan.py qwen2-vl-2b 252 > /dev/null 2>> $OUT/${TAG}_pmc_decode.err
echo "tcp2 pass done"
python3 - <<PY > $OUT/${TAG}_pmc_decode.txt
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter()); dur = collections.defaultdict(lambda: collections.Counter())
for d in ("/tmp/pd_sq", "/tmp/pd_tcp", "/tmp/pd_tcp2"):
    for fn in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "gemm_stream" not in k and "gemm_skinny" not in k:
                continue
            import re
            m = re.search(r"(gemm_\w+_kernel<[^>]*>)", k)
            key = (m.group(1) if m else k[:80]) + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
            dur[key][r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for key, c in sorted(agg.items()):
    print(key)
    for name, v in sorted(c.items()):
        print(f"    {name:44s} {v / n[key][name]:16.0f} per launch   ({n[key][name]} launches, {dur[key][name] / n[key][name] / 1e3:7.1f} us each under the profiler)")
    w = c.get("SQ_WAVE_CYCLES", 0)
    if c.get("TCP_TCC_READ_REQ_sum"):
        print("    L1 -> L2 read requests: mean latency %.0f cycles; the L1 stalled on its pending-request limit for %.0f cycles per CU and launch" % (
            c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"], c.get("TCP_PENDING_STALL_CYCLES_sum", 0) / n[key]["TCP_PENDING_STALL_CYCLES_sum"] / 256 if n[key]["TCP_PENDING_STALL_CYCLES_sum"] else 0))
    if w:
        print("    share of wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %% (of which LDS %.1f %%), issuing %.1f %%" % (
            100 * c.get("SQ_WAIT_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_LDS", 0) / w, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w))
PY
cat $OUT/${TAG}_pmc_decode.txt
rm -rf /tmp/pd_sq /tmp/pd_tcp /tmp/pd_tcp2
