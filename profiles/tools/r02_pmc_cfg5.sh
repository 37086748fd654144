#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
B="--no-cpu --no-extra --windows 1 --roofline-samples 0"
for cn in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cn --output-format csv -d $O/p5_$cn -o p -- python3 bench.py $B --workload matcomp50000 --times-log-rank 5.5 --steps 4 --warmup 1 > $O/p5_$cn.log 2>&1
done
python profiles/pmc_summary.py $(ls $O/p5_FETCH_SIZE/*counter_collection.csv | head -1) $(ls $O/p5_WRITE_SIZE/*counter_collection.csv | head -1) $O/pmc_matcomp50000.json matcomp50000
rm -f $O/p5_*/*counter_collection.csv
python - "$O/pmc_matcomp50000.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("k_op_entry_bip", "k_op_entry", "k_cg_update", "k_cg_dir", "k_spmm2"):
    v = d["kernels"].get(k)
    if v: print(k, v["launches"], round(v["read_bytes_median"] / 1e6, 1), round(v["write_bytes_mean"] / 1e6, 1))
PY
