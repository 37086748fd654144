#!/bin/bash
# perf iteration 3: one-kernel front (k_front_cw + k_wsum): parity tests, then whole iterations with the switch off and on, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p3; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "one_kernel_front or recurrence or carrier or trace_vs_reference_golden or fused_step or constraint_wise" > $O/tests.log 2>&1; tail -5 $O/tests.log
for d in 0 1 0 1; do
  LORADS_FRONT_CW=$d python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_f$d.json 2> $O/bench_f$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_f$d.json').read().strip().splitlines()[-1]); print('front_cw=$d', d['value'], d['ms_per_step_windows'], d['state'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu --no-extra --steps 50 --warmup 5 --windows 1 --roofline-samples 0 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/rand20000_admm_summary.txt; rm -f $T; head -14 $O/rand20000_admm_summary.txt
