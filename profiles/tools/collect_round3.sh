#!/bin/bash
# copies what measure_round3.sh left under gpurun_out/r03m/ into profiles/ (run in the build container after the three gpurun calls)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03m
for w in rand20000 maxcut20000 matcomp50000 blk16x4000; do
  cp $O/${w}_admm_part_summary.txt profiles/r03_${w}_admm_part_summary.txt
  cp $O/${w}_alm_part_summary.txt profiles/r03_${w}_alm_part_summary.txt
  cp $O/${w}_kernel_stats.csv profiles/r03_${w}_kernel_stats.csv
done
cp $O/rand20000_general_form_admm_part_summary.txt profiles/r03_rand20000_general_form_admm_part_summary.txt
cp $O/rand20000_general_form_kernel_stats.csv profiles/r03_general_form_rand20000_kernel_stats.csv
cp $O/pmc_rand20000.json profiles/r03_pmc_rand20000.json
cp $O/pmc_maxcut20000.json profiles/r03_pmc_maxcut20000.json
[ -f $O/pmc_blk16x4000.json ] && cp $O/pmc_blk16x4000.json profiles/r03_pmc_blk16x4000.json
[ -f $O/pmc_matcomp50000.json ] && cp $O/pmc_matcomp50000.json profiles/r03_pmc_matcomp50000.json
cp $O/pmc_rand20000_general_form.json profiles/r03_pmc_rand20000_general_form.json
cp $O/l2_hit_rate_rand20000.json profiles/r03_l2_hit_rate_rand20000.json
cp $O/ubench.txt profiles/r03_ubench.txt
cp $O/size_sweep.txt profiles/r03_size_sweep.txt
cp $O/bench_default.json profiles/r03_bench_default.json
cp $O/bench_cfg5.json profiles/r03_bench_cfg5_matcomp50000.json
cp $O/bench_cfg4_1gpu.json profiles/r03_bench_cfg4_blk16x4000_1gpu.json
cp $O/bench_cfg2_maxcut800.json profiles/r03_bench_cfg2_maxcut800.json
cp $O/rehearsal_gpus2_weak_gloo_one_card.json profiles/r03_rehearsal_gpus2_weak_gloo_one_card.json
cp $O/rehearsal_gpus4_strong_gloo_one_card.json profiles/r03_rehearsal_gpus4_strong_blk16x4000_gloo_one_card.json
cp $O/rehearsal_gpus2_weak_gloo_one_card_hook.json profiles/r03_rehearsal_gpus2_weak_gloo_one_card_hook.json
cp $O/rehearsal_gpus4_strong_gloo_one_card_hook.json profiles/r03_rehearsal_gpus4_strong_blk16x4000_gloo_one_card_hook.json
cp $O/r03_stamp.json profiles/r03_stamp.json
python profiles/tools/stamp.py r03
