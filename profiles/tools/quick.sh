set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${2:-q1}; mkdir -p $O
w=${1:-rand20000}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python3 bench.py --no-cpu --no-extra --workload $w --steps 50 --warmup 5 > $O/kt_$w.log 2>&1
T=$(ls $O/kt_$w/*kernel_trace.csv | head -1)
python profiles/trace_summary.py $T > $O/${w}_admm_summary.txt
rm -f $T
cat $O/${w}_admm_summary.txt | head -30
tail -1 $O/kt_$w.log | cut -c1-600
