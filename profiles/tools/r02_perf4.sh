#!/bin/bash
# perf iteration 4: constraint-major contributions, publish_final prefetch, average folded into the last update
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p4; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_abi_edges.py -x -q -m gpu -k "not fullsize and not every_rank and not cfg5 and not large_vs_arpack" > $O/tests.log 2>&1; tail -5 $O/tests.log
for d in 0 1 0 1; do
  LORADS_FOLD_AVG=$d python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_a$d.json 2> $O/bench_a$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_a$d.json').read().strip().splitlines()[-1]); print('fold_avg=$d', d['value'], d['ms_per_step_windows'], d['state'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu --no-extra --steps 50 --warmup 5 --windows 1 --roofline-samples 0 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/rand20000_admm_summary.txt; rm -f $T; head -12 $O/rand20000_admm_summary.txt
