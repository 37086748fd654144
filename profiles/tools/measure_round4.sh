#!/bin/bash
# Round-4 measurement, part $1 = bench | ab | prof | pmc  (each fits one gpurun call); outputs under gpurun_out/r04m/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
B="--no-cpu --no-extra --windows 1 --roofline-samples 0"
case "$1" in
bench)
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
  python bench.py --workload matcomp50000 --steps 40 --warmup 4 --cpu-budget 30 > $O/bench_cfg5_matcomp50000.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
  python bench.py --workload blk16x4000 --times-log-rank 2.0 --steps 100 --warmup 5 > $O/bench_cfg4_blk16x4000_1gpu.json 2> $O/bench_cfg4_1gpu.err; echo "cfg4 rc=$?"
  python bench.py --workload maxcut800 --times-log-rank 2.0 --steps 200 --warmup 10 --no-extra > $O/bench_cfg2_maxcut800.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
  python bench.py --workload blk2x4000 --times-log-rank 2.0 --steps 200 --warmup 10 --no-extra --no-cpu > $O/bench_blk2x4000_a_2cone_shard_of_cfg4.json 2> $O/bench_blk2.err; echo "blk2 rc=$?"
  python bench.py --workload blk16var --times-log-rank 2.0 --steps 100 --warmup 10 --no-extra --no-cpu > $O/bench_blk16var_unequal_cones.json 2> $O/bench_blk16var.err; echo "blk16var rc=$?"
  # one-card rehearsals of the sharded bench (gloo hook, N processes on the one GPU; each line: the weak replicas AND the strong
  # blk16x4000 run as `extra`, ranks_seen, parity_sharded)
  LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/rehearsal_gpus2.json 2> $O/rehearsal_gpus2.err; echo "gpus2 rc=$?"
  LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 4 --steps 30 --warmup 5 --no-cpu > $O/rehearsal_gpus4.json 2> $O/rehearsal_gpus4.err; echo "gpus4 rc=$?"
  python profiles/tools/stamp.py r04 $O
  ;;
ab)
  python profiles/tools/r04_persist_stamps.py maxcut800:2.0 blk16x4000:2.0 maxcut20000:4.0 blk2x4000:2.0 > $O/persist_phase_times.txt 2>&1; echo "stamps rc=$?"
  bash profiles/tools/r04_persist_ab.sh 200 > $O/persist_ab.txt 2>&1; echo "persist ab rc=$?"
  bash profiles/tools/r04_carry_ab.sh 200 > $O/carry_ab.txt 2>&1; echo "carry ab rc=$?"
  bash profiles/tools/r04_lteam_ab.sh > $O/lteam_ab.txt 2>&1; echo "lteam ab rc=$?"
  python profiles/tools/r04_dinf_cost.py > $O/dinf_cost.txt 2>&1; echo "dinf rc=$?"
  bash profiles/tools/r04_common_rank_ab.sh > $O/common_rank_ab.txt 2>&1; echo "common rank ab rc=$?"
  python profiles/tools/stamp.py r04 $O
  ;;
prof)
  export LORADS_BENCH_NO_PHASE1_RERUN=1   # (the trace is cut into phase 1 / ADMM at the last phase-1 kernel)
  for w in rand20000 maxcut20000 matcomp50000 blk16x4000; do
    TL=4.0; [ $w = matcomp50000 ] && TL=5.5; [ $w = blk16x4000 ] && TL=2.0
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python3 bench.py $B --workload $w --times-log-rank $TL --steps 50 --warmup 5 > $O/kt_$w.log 2>&1
    T=$(ls $O/kt_$w/*kernel_trace.csv | head -1)
    python profiles/trace_summary.py $T > $O/${w}_admm_part_summary.txt
    python profiles/trace_summary.py $T alm > $O/${w}_alm_part_summary.txt
    rm -f $T; cp $O/kt_$w/p_kernel_stats.csv $O/${w}_kernel_stats.csv
  done
  python profiles/tools/stamp.py r04 $O
  ;;
pmc)
  export LORADS_BENCH_NO_PHASE1_RERUN=1
  for w in ${PMC_WORKLOADS:-rand20000 maxcut20000 blk16x4000 matcomp50000}; do
    TL=4.0; [ $w = matcomp50000 ] && TL=5.5; [ $w = blk16x4000 ] && TL=2.0
    for cn in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $cn --output-format csv -d $O/p_${cn}_$w -o p -- python3 bench.py $B --workload $w --times-log-rank $TL --steps 6 --warmup 2 > $O/p_${cn}_$w.log 2>&1
    done
    python profiles/pmc_summary.py $(ls $O/p_FETCH_SIZE_$w/*counter_collection.csv | head -1) $(ls $O/p_WRITE_SIZE_$w/*counter_collection.csv | head -1) $O/pmc_$w.json $w
    python profiles/pmc_summary.py $(ls $O/p_FETCH_SIZE_$w/*counter_collection.csv | head -1) $(ls $O/p_WRITE_SIZE_$w/*counter_collection.csv | head -1) $O/pmc_alm_$w.json $w alm > /dev/null
    rm -f $O/p_FETCH_SIZE_$w/*counter_collection.csv $O/p_WRITE_SIZE_$w/*counter_collection.csv
  done
  python profiles/tools/stamp.py r04 $O
  ;;
esac
echo ALLDONE $1
