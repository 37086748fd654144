#!/bin/bash
# device pre-solve: parity with the host construction + the stage timings either way (all four full-size workloads)
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/presolve_stages.txt
timeout -k 10 900 python -m pytest tests/test_presolve_vs_reference.py -m gpu -x -q > gpurun_out/presolve_tests.log 2>&1 || { tail -40 gpurun_out/presolve_tests.log; exit 1; }
tail -3 gpurun_out/presolve_tests.log
for dev in 1 0; do
  echo "== LORADS_DEV_PRESOLVE=$dev" >> gpurun_out/presolve_stages.txt
  LORADS_DEV_PRESOLVE=$dev LORADS_HIP_VERBOSE=2 timeout -k 10 400 python profiles/tools/startup_times.py >> gpurun_out/presolve_stages.txt 2>&1 || { tail -20 gpurun_out/presolve_stages.txt; exit 1; }
done
cat gpurun_out/presolve_stages.txt
