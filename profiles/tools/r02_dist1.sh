#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
B="--no-cpu --no-extra --windows 3 --roofline-samples 0 --steps 200 --warmup 10"
python bench.py $B > $O/single.json 2> $O/single.err
LORADS_FORCE_DIST=1 python bench.py $B > $O/dist1.json 2> $O/dist1.err
for f in single dist1; do python -c "import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], d['config']['parallelism'])"; done
LORADS_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu --no-extra --windows 1 --roofline-samples 0 --steps 50 --warmup 5 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/dist1_admm_summary.txt; rm -f $T; head -16 $O/dist1_admm_summary.txt; tail -3 $O/dist1_admm_summary.txt
