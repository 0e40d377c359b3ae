#!/bin/bash
# pages per tower launch against the 256-workgroup rounds of its GEMMs: bench.py at several --vit-batch values, interleaved
#   bash tools/ab_vit_batch.sh 12 15 10 12 15
set -e
mkdir -p gpurun_out
for vb in "$@"; do
  python bench.py --no-cpu-baseline --no-extras --vit-batch "$vb" > gpurun_out/ab_vb_$vb.$(date +%s).json 2> gpurun_out/ab_vb.err
  f=$(ls -t gpurun_out/ab_vb_$vb.*.json | head -1)
  python - "$f" "$vb" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"vit_batch {sys.argv[2]:>3}: {d['value']:.3f} {d['unit']}  ms/step {d['ms_per_step']:.1f}  roofline.frac {d['roofline']['frac']:.3f}", flush=True)
PY
done
