#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d4; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "densec or densea or denseac or theta or every_rank_shape" > $O/tests.log 2>&1; tail -2 $O/tests.log
for d in 0 1; do
  LORADS_DENSE_REM=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$d -o p -- python3 profiles/bench_dense_cx.py > $O/dense$d.log 2>&1
  rm -f $O/kt$d/*kernel_trace.csv; grep "cal_obj" $O/dense$d.log
  python - "$O/kt$d/p_kernel_stats.csv" "rem=$d" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_cx" in r["Name"] or "sum_slabs" in r["Name"]:
        print(sys.argv[2], re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:40], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
cp $O/kt1/p_kernel_stats.csv $O/dense_cx_n4096_kernel_stats.csv
