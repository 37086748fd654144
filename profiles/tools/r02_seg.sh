#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02seg; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -s -k "last_workgroup or lockstep_sweep_equals or fullsize_trace or fused_front_of_maxcut" > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; grep -E "CG iterations|passed|failed|^E " $O/pytest.log | cut -c1-250 | tail
[ $rc = 0 ] || exit 1
for cy in 1 0; do
  LORADS_SEG_CARRY=$cy timeout -k 10 300 python bench.py --no-cpu --no-extra --workload blk16x4000 --times-log-rank 2.0 --steps 100 --warmup 5 > $O/cfg4_cy$cy.json 2> $O/cfg4_cy$cy.err || exit 1
done
python - <<'PY'
import json
for cy in (1, 0):
    d = json.loads(open("gpurun_out/r02seg/cfg4_cy%d.json" % cy).read().strip().splitlines()[-1])
    print("carry", cy, round(d["value"], 1), d["unit"], d["ms_per_step_windows"], d["state"])
PY
