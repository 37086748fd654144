#!/bin/bash
# copies what measure_round2.sh (bench, prof, pmc) and r02_pmc_cfg5.sh left under gpurun_out/r02m/ into profiles/ under the round's names
set -e
cd "$(dirname "$0")/../.."
S=gpurun_out/r02m; P=profiles
for w in rand20000 maxcut20000 matcomp50000 blk16x4000; do
  [ -f $S/${w}_admm_part_summary.txt ] || continue
  cp $S/${w}_admm_part_summary.txt $P/r02_${w}_admm_part_summary.txt
  cp $S/${w}_alm_part_summary.txt $P/r02_${w}_alm_part_summary.txt
  cp $S/${w}_kernel_stats.csv $P/r02_${w}_kernel_stats.csv
done
cp $S/rand20000_general_form_admm_part_summary.txt $P/r02_rand20000_general_form_admm_part_summary.txt
cp $S/rand20000_general_form_kernel_stats.csv $P/r02_general_form_rand20000_kernel_stats.csv
cp $S/pmc_rand20000.json $P/r02_pmc_rand20000_iteration.json
cp $S/pmc_rand20000_general_form.json $P/r02_pmc_rand20000.json
cp $S/pmc_maxcut20000.json $P/r02_pmc_maxcut20000.json
cp $S/pmc_matcomp50000.json $P/r02_pmc_matcomp50000.json
cp $S/l2_hit_rate_rand20000.json $P/r02_l2_hit_rate_rand20000.json
cp $S/bench_default.json $P/r02_bench_default.json
cp $S/bench_cfg5.json $P/r02_bench_cfg5_matcomp50000.json
cp $S/bench_cfg4_1gpu.json $P/r02_bench_cfg4_blk16x4000_1gpu.json
cp $S/rehearsal_gpus2_weak_gloo_one_card.json $P/r02_rehearsal_gpus2_weak_gloo_one_card.json
cp $S/rehearsal_gpus4_strong_gloo_one_card.json $P/r02_rehearsal_gpus4_strong_blk16x4000_gloo_one_card.json
cp $S/ubench.txt $P/r02_ubench.txt
cp $S/r02_stamp.json $P/r02_stamp.json
python - <<'PY'
import json
for f in ["r02_bench_default", "r02_bench_cfg5_matcomp50000", "r02_bench_cfg4_blk16x4000_1gpu", "r02_rehearsal_gpus2_weak_gloo_one_card",
          "r02_rehearsal_gpus4_strong_blk16x4000_gloo_one_card"]:
    for ln in open("profiles/%s.json" % f).read().strip().splitlines():
        try:
            d = json.loads(ln)
        except Exception:
            continue
        if isinstance(d, dict) and "value" in d:
            r = d.get("roofline", {})
            print(f, round(d["value"], 1), d["unit"], "| cg/s", round(d.get("cg_iters_per_s", 0)), "| frac", r.get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
                  "|", d["config"]["parallelism"])
PY
