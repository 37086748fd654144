#!/bin/bash
# perf iteration 6: Max-Cut-type cones: front without k_sval, one-kernel evaluation head
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p6; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_abi_edges.py tests/test_multirank_hip.py -x -q -m gpu -k "not fullsize and not every_rank and not cfg5 and not large_vs_arpack" > $O/tests.log 2>&1; tail -5 $O/tests.log
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do set -- $cfg
  LORADS_FRONT_DIAG=$1 LORADS_EVAL_DIAG=$2 python bench.py --workload maxcut20000 --no-cpu --no-extra --steps 200 --warmup 10 --windows 3 --roofline-samples 0 > $O/bench_mc_$1$2.json 2> $O/bench_mc_$1$2.err
  python -c "import json,sys; d=json.loads(open('$O/bench_mc_$1$2.json').read().strip().splitlines()[-1]); print('maxcut front_diag=$1 eval_diag=$2', round(d['value'],1), [round(x,4) for x in d['ms_per_step_windows']], 'cg/s', round(d['cg_iters_per_s']), 'misses', d['speculation_misses_in_timed_region'], d['state'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --workload maxcut20000 --no-cpu --no-extra --steps 50 --warmup 5 --windows 1 --roofline-samples 0 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/maxcut20000_admm_summary.txt; rm -f $T; head -14 $O/maxcut20000_admm_summary.txt
python profiles/tools/ubench.py 200 1,2,30,31,32 > $O/ubench_events.txt 2>$O/ubench.err; cat $O/ubench_events.txt
