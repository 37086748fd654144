#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p9; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_multirank_hip.py -x -q -m gpu -k "lockstep or mix4 or blk4x60 or multirank or ranks or fused_front_of_maxcut or carrier or fused_step" > $O/tests.log 2>&1; tail -3 $O/tests.log
for cfg in "0 0" "1 1" "0 0" "1 1"; do set -- $cfg
LORADS_FRONT_DIAG=$1 LORADS_EVAL_DIAG=$2 python bench.py --workload blk16x4000 --times-log-rank 2.0 --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_$1$2.json 2> $O/bench_$1$2.err
python -c "import json; d=json.loads(open('$O/bench_$1$2.json').read().strip().splitlines()[-1]); print('cfg4 front_diag=$1 eval_diag=$2', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], round(d['cg_iters_per_s']), d['state'])"
done
