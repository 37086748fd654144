#!/bin/bash
# copies what profiles/tools/measure_round4.sh left under gpurun_out/r04m/ into profiles/ (r04_*), run in the build container
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r04m
for f in bench_default bench_cfg5_matcomp50000 bench_cfg4_blk16x4000_1gpu bench_cfg2_maxcut800 bench_blk2x4000_a_2cone_shard_of_cfg4 bench_blk16var_unequal_cones; do cp $O/$f.json profiles/r04_$f.json; done
cp $O/rehearsal_gpus2.json profiles/r04_rehearsal_gpus2.json
cp $O/rehearsal_gpus4.json profiles/r04_rehearsal_gpus4.json
for w in rand20000 maxcut20000 matcomp50000 blk16x4000; do
  cp $O/${w}_admm_part_summary.txt profiles/r04_${w}_admm_part_summary.txt
  cp $O/${w}_alm_part_summary.txt profiles/r04_${w}_alm_part_summary.txt
  cp $O/${w}_kernel_stats.csv profiles/r04_${w}_kernel_stats.csv
  cp $O/pmc_$w.json profiles/r04_pmc_$w.json
done
for w in rand20000 matcomp50000; do [ -f $O/pmc_alm_$w.json ] && cp $O/pmc_alm_$w.json profiles/r04_pmc_alm_$w.json; done
cp $O/persist_phase_times.txt profiles/r04_persist_phase_times.txt
cp $O/persist_ab.txt profiles/r04_persist_ab.txt
cp $O/dinf_cost.txt profiles/r04_dinf_cost.txt
(head -1 profiles/r04_carry_ab.txt; cat $O/carry_ab.txt) > /tmp/c.txt && mv /tmp/c.txt profiles/r04_carry_ab.txt
(head -3 profiles/r04_lteam_ab.txt; cat $O/lteam_ab.txt) > /tmp/lt.txt && mv /tmp/lt.txt profiles/r04_lteam_ab.txt
(head -1 profiles/r04_common_rank_ab.txt; cat $O/common_rank_ab.txt) > /tmp/cr.txt && mv /tmp/cr.txt profiles/r04_common_rank_ab.txt
cp $O/r04_stamp.json profiles/r04_stamp.json
echo collected
