#!/bin/bash
# separable shards vs the m-vector form: the sharded tests, then the one-card rehearsals of bench.py --gpus N in both forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02sep; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_multirank_hip.py -x -q -s > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc = 0 ] || exit 1
export LORADS_DIST_BACKEND=gloo LORADS_FORCE_DEVICE=0
for sep in 1 0; do
  LORADS_SEPARABLE=$sep timeout -k 10 300 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu > $O/gpus2_weak_sep$sep.json 2> $O/gpus2_weak_sep$sep.err || exit 1
  LORADS_SEPARABLE=$sep timeout -k 10 300 python bench.py --gpus 4 --scaling strong --steps 50 --warmup 5 --no-cpu > $O/gpus4_strong_sep$sep.json 2> $O/gpus4_strong_sep$sep.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02sep/*.json")):
    for ln in open(f):
        try: d = json.loads(ln)
        except Exception: continue
        print(f.split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), d["config"]["parallelism"], d["state"])
PY
