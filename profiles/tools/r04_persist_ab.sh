#!/bin/bash
# A/B of the one-launch ADMM iteration (persist.inc) against the launch-by-launch form on the Max-Cut-type workloads.
# usage (GPU box): bash profiles/tools/r04_persist_ab.sh [steps]
set -e
STEPS=${1:-200}
mkdir -p gpurun_out/r04_ab
for wl in maxcut800:2.0 blk16x4000:2.0 maxcut20000:4.0 blk16var:2.0 maxcut4000:2.0; do
  name=${wl%%:*}; tlr=${wl##*:}
  for p in 1 0; do
    LORADS_PERSIST=$p python bench.py --workload $name --times-log-rank $tlr --steps $STEPS --warmup 20 --no-cpu --no-extra --roofline-samples 0 --windows 3 \
      > gpurun_out/r04_ab/${name}_persist$p.json 2> gpurun_out/r04_ab/${name}_persist$p.log
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_ab/${name}_persist$p.json"))
print("${name} persist=$p: %.1f ADMM it/s, %.4f ms/step (windows %s), %.0f CG it/s, %.2f CG/it" % (d["value"], d["ms_per_step"], ["%.4f"%x for x in d["ms_per_step_windows"]], d["cg_iters_per_s"], d["cg_iters_per_admm_iter"]))
PY
  done
done
