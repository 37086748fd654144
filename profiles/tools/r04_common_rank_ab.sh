#!/bin/bash
# LORADS_COMMON_RANK (csrc/hip/build.inc: the cones of a block-separable context share the largest device rank, zero columns beyond
# their own) against per-cone ranks: phase 1 and ADMM of contexts with unequal cones.  usage (GPU box): bash profiles/tools/r04_common_rank_ab.sh
set -e
mkdir -p gpurun_out/r04_cr
for wl in blk16var randblk8var; do
  for p in 1 0; do
    LORADS_COMMON_RANK=$p LORADS_BENCH_NO_PHASE1_RERUN=1 python bench.py --workload $wl --times-log-rank 2.0 --steps 100 --warmup 10 --no-cpu --no-extra --roofline-samples 0 --windows 1 \
      > gpurun_out/r04_cr/${wl}_cr$p.json 2> gpurun_out/r04_cr/${wl}_cr$p.log
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_cr/${wl}_cr$p.json"))
p=d["phase1"]
print("${wl} common rank=$p: ADMM %.1f it/s (%.4f ms/step, %.1f launches), %.0f CG it/s; phase 1 %d inner its, %.1f us and %.1f launches per inner iteration; %s" % (d["value"], d["ms_per_step"], d["launches_per_step"], d["cg_iters_per_s"], p["inner_iters"], p["us_per_inner_iter"], p["launches_per_inner_iter"], d["config"]["workload"][:60]))
PY
  done
done
