#!/bin/bash
# A/B of the carried objective gather of the one-launch ADMM iteration (LORADS_PERSIST_CARRY, persist.inc): two gathers per iteration
# instead of three.  usage (GPU box): bash profiles/tools/r04_carry_ab.sh [steps]
set -e
STEPS=${1:-200}
mkdir -p gpurun_out/r04_carry
for wl in maxcut800:2.0 blk16x4000:2.0 maxcut20000:4.0 blk16var:2.0 blk2x4000:2.0; do
  name=${wl%%:*}; tlr=${wl##*:}
  for p in 1 0; do
    LORADS_PERSIST_CARRY=$p python bench.py --workload $name --times-log-rank $tlr --steps $STEPS --warmup 20 --no-cpu --no-extra --roofline-samples 0 --windows 3 \
      > gpurun_out/r04_carry/${name}_carry$p.json 2> gpurun_out/r04_carry/${name}_carry$p.log
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_carry/${name}_carry$p.json"))
print("${name} carry=$p: %.1f ADMM it/s, %.4f ms/step (windows %s), %.0f CG it/s, %.2f CG/it, pObj %.12g" % (d["value"], d["ms_per_step"], ["%.4f"%x for x in d["ms_per_step_windows"]], d["cg_iters_per_s"], d["cg_iters_per_admm_iter"], d["state"]["pObj"]))
PY
  done
done
