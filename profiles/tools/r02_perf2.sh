#!/bin/bash
# perf iteration 2: non-temporal streams in the gather kernels (LORADS_DEEP_GATHER switch), in isolation and in whole iterations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p2; mkdir -p $O
python profiles/tools/ubench.py 200 2,12,4,15,5,16 > $O/ubench_events.txt 2>$O/ubench.err; cat $O/ubench_events.txt
for d in 0 1 0 1; do
  LORADS_DEEP_GATHER=$d python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_nt$d.json 2> $O/bench_nt$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_nt$d.json').read().strip().splitlines()[-1]); print('nt=$d', d['value'], d['ms_per_step_windows'])"
done
for w in maxcut20000 matcomp50000; do for d in 0 1; do
  LORADS_DEEP_GATHER=$d python bench.py --workload $w --no-cpu --no-extra --steps 100 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_${w}_nt$d.json 2> $O/bench_${w}_nt$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_${w}_nt$d.json').read().strip().splitlines()[-1]); print('$w nt=$d', d['value'], d['ms_per_step_windows'])"
done; done
