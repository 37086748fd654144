#!/bin/bash
# Round-3 measurement, part $1 = bench | prof | pmc  (each fits one gpurun call); outputs under gpurun_out/r03m/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
B="--no-cpu --no-extra --windows 1 --roofline-samples 0"
case "$1" in
bench)
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
  python bench.py --workload matcomp50000 --steps 40 --warmup 4 --cpu-budget 30 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
  python bench.py --workload blk16x4000 --times-log-rank 2.0 --steps 100 --warmup 5 > $O/bench_cfg4_1gpu.json 2> $O/bench_cfg4_1gpu.err; echo "cfg4 rc=$?"
  python bench.py --workload maxcut800 --times-log-rank 2.0 --steps 200 --warmup 10 --no-extra > $O/bench_cfg2_maxcut800.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"
  # one-card rehearsals of the sharded bench (gloo hook, N processes on the one GPU): the evaluation's scalars from host to host through
  # shared memory (the default on one node) and, _hook, through the all-reduce hook (LORADS_SHM_EXCHANGE=0)
  LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/rehearsal_gpus2_weak_gloo_one_card.json 2> $O/rehearsal_gpus2.err; echo "gpus2 rc=$?"
  LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 4 --scaling strong --steps 50 --warmup 5 --no-cpu > $O/rehearsal_gpus4_strong_gloo_one_card.json 2> $O/rehearsal_gpus4.err; echo "gpus4 rc=$?"
  LORADS_SHM_EXCHANGE=0 LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/rehearsal_gpus2_weak_gloo_one_card_hook.json 2> $O/rehearsal_gpus2_hook.err; echo "gpus2 hook rc=$?"
  LORADS_SHM_EXCHANGE=0 LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0 python bench.py --gpus 4 --scaling strong --steps 50 --warmup 5 --no-cpu > $O/rehearsal_gpus4_strong_gloo_one_card_hook.json 2> $O/rehearsal_gpus4_hook.err; echo "gpus4 hook rc=$?"
  python profiles/tools/ubench.py 200 1,2,30,31,32,8,10,23,24 > $O/ubench.txt 2> $O/ubench.err
  python profiles/tools/size_sweep.py 200 > $O/size_sweep.txt 2> $O/size_sweep.err
  python profiles/tools/stamp.py r03 $O
  ;;
prof)
  for w in rand20000 maxcut20000 matcomp50000 blk16x4000; do
    TL=4.0; [ $w = matcomp50000 ] && TL=5.5; [ $w = blk16x4000 ] && TL=2.0
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python3 bench.py $B --workload $w --times-log-rank $TL --steps 50 --warmup 5 > $O/kt_$w.log 2>&1
    T=$(ls $O/kt_$w/*kernel_trace.csv | head -1)
    python profiles/trace_summary.py $T > $O/${w}_admm_part_summary.txt
    python profiles/trace_summary.py $T alm > $O/${w}_alm_part_summary.txt
    rm -f $T; cp $O/kt_$w/p_kernel_stats.csv $O/${w}_kernel_stats.csv
  done
  # the operator's general form (k_cw + k_spmm_ell every application), as the roofline pass of bench.py times it
  LORADS_FRONT_CW=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_general -o p -- python3 bench.py $B --steps 50 --warmup 5 > $O/kt_general.log 2>&1
  T=$(ls $O/kt_general/*kernel_trace.csv | head -1); python profiles/trace_summary.py $T > $O/rand20000_general_form_admm_part_summary.txt; rm -f $T
  cp $O/kt_general/p_kernel_stats.csv $O/rand20000_general_form_kernel_stats.csv
  python profiles/tools/stamp.py r03 $O
  ;;
pmc)
  for w in ${PMC_WORKLOADS:-rand20000 maxcut20000 blk16x4000 matcomp50000}; do
    TL=4.0; [ $w = matcomp50000 ] && TL=5.5; [ $w = blk16x4000 ] && TL=2.0
    for cn in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $cn --output-format csv -d $O/p_${cn}_$w -o p -- python3 bench.py $B --workload $w --times-log-rank $TL --steps 6 --warmup 2 > $O/p_${cn}_$w.log 2>&1
    done
    python profiles/pmc_summary.py $(ls $O/p_FETCH_SIZE_$w/*counter_collection.csv | head -1) $(ls $O/p_WRITE_SIZE_$w/*counter_collection.csv | head -1) $O/pmc_$w.json $w
    rm -f $O/p_FETCH_SIZE_$w/*counter_collection.csv $O/p_WRITE_SIZE_$w/*counter_collection.csv
  done
  for cn in FETCH_SIZE WRITE_SIZE; do
    LORADS_FRONT_CW=0 rocprofv3 --pmc $cn --output-format csv -d $O/pg_$cn -o p -- python3 bench.py $B --steps 6 --warmup 2 > $O/pg_$cn.log 2>&1
  done
  python profiles/pmc_summary.py $(ls $O/pg_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/pg_WRITE_SIZE/*counter_collection.csv | head -1) $O/pmc_rand20000_general_form.json rand20000
  rm -f $O/pg_*/*counter_collection.csv
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o p -- python3 bench.py $B --steps 6 --warmup 2 > $O/l2.log 2>&1
  python - "$O" <<'PY'
import collections, csv, glob, json, re, sys
O = sys.argv[1]
f = glob.glob(O + "/l2/*counter_collection.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    last = max(i for i, r in enumerate(rows) if "k_his_two" in r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows[last + 1:]:
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {k: {"TCC_HIT_sum": v.get("TCC_HIT_sum", 0.0), "TCC_MISS_sum": v.get("TCC_MISS_sum", 0.0),
               "l2_hit_rate": v.get("TCC_HIT_sum", 0.0) / max(1.0, v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))} for k, v in agg.items()}
    json.dump(out, open(O + "/l2_hit_rate_rand20000.json", "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda x: -x[1]["TCC_MISS_sum"])[:10]: print(k, round(v["l2_hit_rate"], 3))
import os
for x in f: os.remove(x)
PY
  python profiles/tools/stamp.py r03 $O
  ;;
esac
echo ALLDONE $1
