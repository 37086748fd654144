set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r01g}; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in rand20000 maxcut20000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python3 bench.py --no-cpu --no-extra --workload $w --steps 50 --warmup 5 > $O/kt_$w.log 2>&1
  T=$(ls $O/kt_$w/*kernel_trace.csv | head -1)
  python profiles/trace_summary.py $T > $O/${w}_admm_summary.txt
  python profiles/trace_summary.py $T alm > $O/${w}_alm_summary.txt
  rm -f $T
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_$w -o p -- python3 bench.py --no-cpu --no-extra --workload $w --steps 6 --warmup 2 > $O/pf_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw_$w -o p -- python3 bench.py --no-cpu --no-extra --workload $w --steps 6 --warmup 2 > $O/pw_$w.log 2>&1
  python profiles/pmc_summary.py $(ls $O/pf_$w/*counter_collection.csv | head -1) $(ls $O/pw_$w/*counter_collection.csv | head -1) $O/pmc_$w.json $w
  rm -f $O/pf_$w/*counter_collection.csv $O/pw_$w/*counter_collection.csv
done
echo ALLDONE
