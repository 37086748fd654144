#!/bin/bash
# round-2 second GPU call: reference loops through the shim, whole-solve parity report, cfg5 full-size trace, cfg5 bench parity
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
python -m pytest tests/test_reference_shim.py -q -m gpu -s > $O/test_shim.log 2>&1; echo "shim test rc=$?" | tee -a $O/summary.txt
python profiles/tools/solve_parity_report.py > $O/solve_parity.txt 2> $O/solve_parity.err; echo "solve report rc=$?" | tee -a $O/summary.txt
python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "fullsize_trace and matcomp50000" > $O/test_cfg5_trace.log 2>&1; echo "cfg5 trace rc=$?" | tee -a $O/summary.txt
python bench.py --workload matcomp50000 --steps 20 --warmup 2 --cpu-budget 30 --windows 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "bench cfg5 rc=$?" | tee -a $O/summary.txt
tail -n 30 $O/test_shim.log; cat $O/solve_parity.txt; tail -n 5 $O/test_cfg5_trace.log
