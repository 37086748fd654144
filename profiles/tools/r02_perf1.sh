#!/bin/bash
# perf iteration 1: deep-gather variants in isolation (ubench), then whole iterations with the switch off and on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p1; mkdir -p $O
python profiles/tools/ubench.py 200 0,1,2,12,4,15,5,16,8,10,11 > $O/ubench_events.txt 2>$O/ubench.err; cat $O/ubench_events.txt
for d in 0 1; do
  LORADS_DEEP_GATHER=$d python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_deep$d.json 2> $O/bench_deep$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_deep$d.json').read().strip().splitlines()[-1]); print('deep=$d', d['value'], d['ms_per_step_windows'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu --no-extra --steps 50 --warmup 5 --windows 1 --roofline-samples 0 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/rand20000_admm_summary.txt; rm -f $T; cat $O/rand20000_admm_summary.txt
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "trace_vs_reference_golden or carrier or recurrence" > $O/tests.log 2>&1; tail -3 $O/tests.log
