#!/bin/bash
# A/B of one environment switch in ONE gpurun call (same box, same clocks): tools/ab_env.sh NAME "bench args" -> gpurun_out/ab_NAME_{1,0}.json
# The kernel-path switches exist in the diagnostic library only (-DHWOCR_DIAG): HWOCR_DIAG_LIB=1 makes _lib.hip() build and load it.
set -e
export HWOCR_DIAG_LIB=1
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
for round in a b; do
  for m in 1 0; do
    env $1=$m python3 $R/bench.py $2 > $R/gpurun_out/ab_$1_${m}_$round.json 2> $R/gpurun_out/ab_$1_${m}_$round.err
    python3 -c "
import json,sys; d=json.load(open('$R/gpurun_out/ab_$1_${m}_$round.json')); print('$1=$m', '$round', round(d['value'],3), {k: round(v,1) for k,v in d['phases_ms_per_step'].items() if isinstance(v,(int,float))})"
  done
done
