#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) for cfg4 and cfg5 -- the same recipe as measure_round3.sh pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
B="--no-cpu --no-extra --windows 1 --roofline-samples 0"
for w in blk16x4000 matcomp50000; do
  TL=5.5; [ $w = blk16x4000 ] && TL=2.0
  for cn in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $cn --output-format csv -d $O/p_${cn}_$w -o p -- python3 bench.py $B --workload $w --times-log-rank $TL --steps 6 --warmup 2 > $O/p_${cn}_$w.log 2>&1 || { echo "$w $cn failed"; tail -5 $O/p_${cn}_$w.log; exit 1; }
    echo "$w $cn done"
  done
  python profiles/pmc_summary.py $(ls $O/p_FETCH_SIZE_$w/*counter_collection.csv | head -1) $(ls $O/p_WRITE_SIZE_$w/*counter_collection.csv | head -1) $O/pmc_$w.json $w
  rm -f $O/p_FETCH_SIZE_$w/*counter_collection.csv $O/p_WRITE_SIZE_$w/*counter_collection.csv
done
