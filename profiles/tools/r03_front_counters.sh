#!/bin/bash
# Round 3: counter evidence for k_front_cw against the front-shaped gather probes.
# Separate --pmc passes (SQ / TCP / TCC / GRBM slots), no trace domains beside them; outputs under gpurun_out/r03c/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
V=${1:-30,31,23,24,2,1}
rocprofv3 -L > $O/counters_avail.txt 2>&1
pass() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o p -- python3 profiles/tools/ubench.py 20 $V > $O/$n.log 2>&1
  echo "pass $n rc=$?"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU
pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 profiles/tools/ubench.py 20 $V > $O/kt.log 2>&1
echo "kt rc=$?"
python3 profiles/tools/r03_counter_summary.py $O > $O/front_cw_counters.json
echo ALLDONE
