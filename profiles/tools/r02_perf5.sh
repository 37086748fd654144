#!/bin/bash
# perf iteration 5: speculation window on Max-Cut / headline / cfg4 / cfg5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p5; mkdir -p $O
for w in maxcut20000 rand20000 blk16x4000 matcomp50000; do for d in 1 4; do
  TL=4.0; [ $w = blk16x4000 ] && TL=2.0; [ $w = matcomp50000 ] && TL=5.5
  LORADS_SPEC_WINDOW=$d python bench.py --workload $w --times-log-rank $TL --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_${w}_w$d.json 2> $O/bench_${w}_w$d.err
  python -c "import json,sys; d=json.loads(open('$O/bench_${w}_w$d.json').read().strip().splitlines()[-1]); print('$w window=$d', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], 'cg/s', round(d['cg_iters_per_s']), 'misses', d['speculation_misses_in_timed_region'])"
done; done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused_step or lockstep or carrier" > $O/tests.log 2>&1; tail -3 $O/tests.log
