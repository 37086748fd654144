#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c4; mkdir -p $O
B="--no-cpu --no-extra --windows 1 --roofline-samples 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py $B --workload blk16x4000 --times-log-rank 2.0 --steps 50 --warmup 5 > $O/kt.log 2>&1
T=$(ls $O/kt/*kernel_trace.csv | head -1)
python profiles/trace_summary.py $T > $O/blk16x4000_admm_part_summary.txt
rm -f $T
head -30 $O/blk16x4000_admm_part_summary.txt
