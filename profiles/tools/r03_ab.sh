#!/bin/bash
# interleaved A/B of one environment switch on bench.py's timed region: $1 = VAR, $2 = value A, $3 = value B, $4.. = bench args
V=$1; A=$2; Bv=$3; shift 3
B="--no-cpu --no-extra --roofline-samples 0 --windows 5"
mkdir -p gpurun_out
for rep in 1 2 3; do
  for val in $A $Bv; do
    env $V=$val python bench.py $B "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$V=$val', 'value %.1f' % d['value'], 'windows', ['%.4f' % x for x in d.get('ms_per_step_windows',[])], flush=True)" || exit 1
  done
done
