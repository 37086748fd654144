#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02bip; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -s -k "bipartite or matcomp or cfg5 or every_rank_shape or test_trace_vs_reference_golden" > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; grep -E "CG iterations|passed|failed|^E " $O/pytest.log | cut -c1-250 | tail
[ $rc = 0 ] || exit 1
for on in 1 0; do
  LORADS_ENTRY_BIP=$on timeout -k 10 300 python profiles/tools/ubench.py 100 100 matcomp50000 5.5 > $O/ubench_bip$on.txt 2>&1
  LORADS_ENTRY_BIP=$on timeout -k 10 400 python bench.py --no-cpu --no-extra --workload matcomp50000 --steps 40 --warmup 4 > $O/cfg5_bip$on.json 2> $O/cfg5_bip$on.err || exit 1
done
tail -2 $O/ubench_bip1.txt $O/ubench_bip0.txt
python - <<'PY'
import json
for on in (1, 0):
    d = json.loads(open("gpurun_out/r02bip/cfg5_bip%d.json" % on).read().strip().splitlines()[-1])
    print("bip", on, round(d["value"], 1), d["unit"], d["ms_per_step_windows"], round(d["cg_iters_per_s"]), d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["state"])
PY
