#!/bin/bash
# Round 4: counter evidence for the one-launch ADMM iteration k_admm_diag (cfg3a Max-Cut n = 20000, cfg4 blk16x4000) and for phase 1's
# k_lbfgs_team.  Separate --pmc passes (SQ / TCP / TCC slots), no trace domains beside them; outputs under gpurun_out/r04c/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LORADS_BENCH_NO_PHASE1_RERUN=1
O=gpurun_out/r04c; mkdir -p $O
B="--no-cpu --no-extra --windows 0 --roofline-samples 0 --steps 8 --warmup 2"
for w in maxcut20000 blk16x4000; do
  TL=4.0; [ $w = blk16x4000 ] && TL=2.0
  pass() { # name, counters...
    n=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $O/${w}_$n -o p -- python3 bench.py $B --workload $w --times-log-rank $TL > $O/${w}_$n.log 2>&1
    echo "$w pass $n rc=$?"
  }
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU
  pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT
  pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
  pass tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
python3 - "$O" > $O/admm_diag_counters.json <<'PY'
import collections, csv, glob, json, re, sys
O = sys.argv[1]
out = {}
for w in ("maxcut20000", "blk16x4000"):
    res = collections.defaultdict(dict)
    for d in ("sq1", "sq2", "tcp", "tcc"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, w, d), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                m = re.search(r"(k_admm_diag<[^>]*>|k_lbfgs_team<[^>]*>)", r["Kernel_Name"])
                if m:
                    agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    for k in ("VGPR_Count", "Arch_VGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                        if k in r and r[k]:
                            res[m.group(1)][k] = r[k]
            for k, cs in agg.items():
                for cn, v in cs.items():
                    res[k][cn] = sum(v) / len(v)
                    res[k]["launches"] = len(v)
    out[w] = res
json.dump(out, sys.stdout, indent=1, sort_keys=True)
PY
find $O -name "*counter_collection.csv" -delete
echo ALLDONE
