#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: rocprofv3 per-kernel summary of the default bench and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only beside them) for the traffic summary.  Only the small
# summaries are left under gpurun_out/ (the raw traces are tens of MB).   usage: tools/profile_round.sh <tag> [bench flags...]
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks /tmp/prof_fetch /tmp/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" \
    > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
cp /tmp/prof_ks/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_bench_default.csv
echo "kernel stats done"
# the same with one batch at a time (--lanes 1): per-launch durations of kernels that have the chip to themselves (with two lanes two
# launches run side by side and each takes about twice as long)
rm -rf /tmp/prof_ks1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks1 -- python3 $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" \
    > $OUT/${TAG}_bench_one_lane_under_rocprof.json 2>> $OUT/${TAG}_rocprof.err
cp /tmp/prof_ks1/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_bench_one_lane.csv
rm -rf /tmp/prof_ks1
echo "one-lane kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $R/bench.py --pages 12 --steps 1 --warmup 0 --new-tokens 4 \
    --no-cpu-baseline --no-extras "$@" > /dev/null 2>> $OUT/${TAG}_rocprof.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $R/bench.py --pages 12 --steps 1 --warmup 0 --new-tokens 4 \
    --no-cpu-baseline --no-extras "$@" > /dev/null 2>> $OUT/${TAG}_rocprof.err
echo "write pass done"
python3 $R/tools/pmc_summary.py /tmp/prof_fetch/*/*_counter_collection.csv /tmp/prof_write/*/*_counter_collection.csv $OUT/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/prof_util -- python3 $R/bench.py --pages 12 --steps 1 --warmup 0 \
    --new-tokens 4 --no-cpu-baseline --no-extras "$@" > /dev/null 2>> $OUT/${TAG}_rocprof.err
python3 $R/tools/pmc_util.py /tmp/prof_util/*/*_counter_collection.csv > $OUT/${TAG}_pmc_util.txt
cat $OUT/${TAG}_pmc_util.txt
rm -rf /tmp/prof_ks /tmp/prof_fetch /tmp/prof_write /tmp/prof_util
