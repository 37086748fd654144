#!/bin/bash
# pairs of doubles per thread and vector (= workgroups of the team) of the one-launch L-BFGS direction: microseconds per inner iteration
set -e
for wl in rand20000 matcomp50000; do
for np in 4 6 8 12; do
LORADS_LBFGS_TEAM_NP=$np python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu --no-extra --roofline-samples 0 --windows 0 > gpurun_out/np_${wl}_$np.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/np_${wl}_$np.json")); p=d["phase1"]
print("$wl np=$np: %d inner, %.1f us per inner iteration" % (p["inner_iters"], p["us_per_inner_iter"]))
PY
done
done
