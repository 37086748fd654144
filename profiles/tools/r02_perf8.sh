#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p8; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_multirank_hip.py -x -q -m gpu -k "not fullsize and not every_rank and not cfg5 and not large_vs_arpack" > $O/tests.log 2>&1; tail -2 $O/tests.log
for d in 0 1 0 1; do
LORADS_FUSE_EVAL=$d python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench$d.json 2> $O/bench$d.err
python -c "import json; d=json.loads(open('$O/bench$d.json').read().strip().splitlines()[-1]); print('fuse_eval=$d', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], d['state'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu --no-extra --steps 50 --warmup 5 --windows 1 --roofline-samples 0 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/rand20000_admm_summary.txt; rm -f $T; head -9 $O/rand20000_admm_summary.txt
