#!/bin/bash
# bench.py at several values of one flag, interleaved in one call:  bash tools/ab_bench_flag.sh --prefill-batch 16 24 16 24
set -e
flag=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  f=gpurun_out/ab$(echo "$flag" | tr - _)_$v.$(date +%s).json
  python bench.py --no-cpu-baseline --no-extras "$flag" "$v" > "$f" 2> gpurun_out/ab_flag.err
  python - "$f" "$flag" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} {sys.argv[3]:>4}: {d['value']:.3f} {d['unit']}  ms/step {d['ms_per_step']:.1f}", flush=True)
PY
done
