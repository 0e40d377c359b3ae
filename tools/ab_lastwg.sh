set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests/test_ops_gpu.py -x -q -m gpu -k "attn_decode or argmax" > $R/gpurun_out/r03t_ops.log 2>&1
for m in 1 0; do
  export HWOCR_DECODE_LASTWG=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1_$m -- python3 $R/bench.py --pages 1 --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r03t_1page_lastwg$m.json 2> /tmp/p1_$m.err
  cp $(find /tmp/p1_$m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03t_kernel_stats_1page_lastwg$m.csv
done
