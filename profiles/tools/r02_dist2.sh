#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s2; mkdir -p $O
python -m pytest tests/test_multirank_hip.py tests/test_hip_parity.py -x -q -m gpu -k "multirank or rccl or ranks or one_kernel_front or carrier or fused_step" > $O/tests.log 2>&1; tail -3 $O/tests.log
B="--no-cpu --no-extra --windows 3 --roofline-samples 0 --steps 200 --warmup 10"
LORADS_FORCE_DIST=1 python bench.py $B > $O/dist1.json 2> $O/dist1.err
python -c "import json; d=json.loads(open('$O/dist1.json').read().strip().splitlines()[-1]); print('dist1', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], d['state'])"
