#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: rocprofv3 per-kernel summary of one batch at a time for another
# configuration + its default (two-lane) bench line.   usage: tools/profile_config.sh <tag> <bench flags...>
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cfg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg -- python3 $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" \
    > $OUT/${TAG}_bench_one_lane_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
cp /tmp/prof_cfg/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_bench_one_lane.csv
rm -rf /tmp/prof_cfg
echo "kernel stats done"
