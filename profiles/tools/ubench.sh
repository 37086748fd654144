set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ub1}; mkdir -p $O
python profiles/tools/ubench.py 200 > $O/ubench_events.txt 2>$O/ubench.err
cat $O/ubench_events.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 profiles/tools/ubench.py 50 > $O/ubench_prof.log 2>&1
rm -f $O/kt/*kernel_trace.csv
python - <<'PY' "$O"
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1] + "/kt/p_kernel_stats.csv")))
for r in rows:
    if int(r["Calls"]) in (53,) or "ubench" in r["Name"]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
        print("%-45s calls %4s avg %8.2f us  min %8.2f" % (n[:45], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
