#!/bin/bash
# round-2 first GPU call: cfg5 oracle test, default bench line, 2-rank rehearsal on one card, cfg5 bench line
set -o pipefail
mkdir -p gpurun_out/r02a
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cfg5_slots" > gpurun_out/r02a/test_cfg5.log 2>&1; echo "cfg5 test rc=$?" | tee -a gpurun_out/r02a/summary.txt
python bench.py --steps 100 --warmup 5 > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err; echo "bench default rc=$?" | tee -a gpurun_out/r02a/summary.txt
LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > gpurun_out/r02a/bench_gpus2_gloo.json 2> gpurun_out/r02a/bench_gpus2_gloo.err; echo "bench gpus2 rc=$?" | tee -a gpurun_out/r02a/summary.txt
LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 4 --scaling strong --steps 50 --warmup 5 --no-cpu > gpurun_out/r02a/bench_strong4_gloo.json 2> gpurun_out/r02a/bench_strong4_gloo.err; echo "bench strong4 rc=$?" | tee -a gpurun_out/r02a/summary.txt
python bench.py --workload matcomp50000 --steps 20 --warmup 2 --cpu-budget 30 > gpurun_out/r02a/bench_cfg5.json 2> gpurun_out/r02a/bench_cfg5.err; echo "bench cfg5 rc=$?" | tee -a gpurun_out/r02a/summary.txt
tail -c 600 gpurun_out/r02a/test_cfg5.log
