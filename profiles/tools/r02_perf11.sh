#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p11; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "one_kernel_front or recurrence or carrier or trace_vs_reference_golden or fused_step" > $O/tests.log 2>&1; tail -2 $O/tests.log
for d in 1 0 1 0; do
if [ $d = 1 ]; then export LORADS_NO_TILE_SORT=1; else unset LORADS_NO_TILE_SORT; fi
python bench.py --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench$d.json 2> $O/bench$d.err
python -c "import json; d=json.loads(open('$O/bench$d.json').read().strip().splitlines()[-1]); print('no_tile_sort=$d', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']])"
python profiles/tools/ubench.py 200 30,31 2>/dev/null
done
