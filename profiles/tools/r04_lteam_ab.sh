#!/bin/bash
# A/B of phase 1's one-launch L-BFGS history update + direction and the shared passes behind it (LORADS_LBFGS_TEAM,
# LORADS_ALM_FUSED_TAIL; lbfgs_team.inc): microseconds per inner iteration from bench.py's phase1 object.
# usage (GPU box): bash profiles/tools/r04_lteam_ab.sh
set -e
mkdir -p gpurun_out/r04_lteam
for wl in rand20000 matcomp50000 maxcut20000; do
  for p in 11 10 00; do
    extra=""
    [ $wl = maxcut20000 ] && extra="--times-log-rank 4.0"
    LORADS_LBFGS_TEAM=${p:0:1} LORADS_ALM_FUSED_TAIL=${p:1:1} python bench.py --workload $wl $extra --steps 50 --warmup 10 --no-cpu --no-extra --roofline-samples 0 --windows 0 \
      > gpurun_out/r04_lteam/${wl}_team$p.json 2> gpurun_out/r04_lteam/${wl}_team$p.log
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_lteam/${wl}_team$p.json"))
p=d["phase1"]
print("${wl} team/tail=$p: phase 1 %d inner iterations (%d outer) in %.4f s: %.1f us per inner iteration; ADMM %.1f it/s" % (p["inner_iters"], p["outer_iters"], p["seconds"], p["us_per_inner_iter"], d["value"]))
PY
  done
done
