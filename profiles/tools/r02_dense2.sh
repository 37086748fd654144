#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d2; mkdir -p $O
for cfg in "4 512" "8 512" "4 1024" "8 1024" "4 2048" "8 256"; do set -- $cfg
  LORADS_DENSE_UN=$1 LORADS_DENSE_WG=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_$2 -o p -- python3 profiles/bench_dense_cx.py > $O/dense_$1_$2.log 2>&1
  rm -f $O/kt_$1_$2/*kernel_trace.csv
  python - "$O/kt_$1_$2/p_kernel_stats.csv" "$1 $2" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_cx" in r["Name"] or "sum_slabs" in r["Name"]:
        print(sys.argv[2], re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:40], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
