#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "densec or theta or every_rank" > $O/tests.log 2>&1; tail -3 $O/tests.log
for d in 0 1; do
  LORADS_DENSE_B=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$d -o p -- python3 profiles/bench_dense_cx.py > $O/dense$d.log 2>&1
  tail -2 $O/dense$d.log; rm -f $O/kt$d/*kernel_trace.csv
  python - "$O/kt$d/p_kernel_stats.csv" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_cx" in r["Name"] or "sum_slabs" in r["Name"]:
        print(re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:60], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
done
N=8192 LORADS_DENSE_B=1 python profiles/bench_dense_cx.py 2>&1 | tail -2
