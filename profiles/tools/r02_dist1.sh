#!/bin/bash
# the sharded forms' own cost on ONE rank (native RCCL hook on a communicator of one, in stream order): plain single GPU, separable, m-vector
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02dist1; mkdir -p $O
B="--no-cpu --no-extra --roofline-samples 0 --steps 200 --warmup 10"
python bench.py $B > $O/plain.json 2> $O/plain.err || exit 1
LORADS_FORCE_DIST=1 python bench.py $B > $O/sep.json 2> $O/sep.err || exit 1
LORADS_FORCE_DIST=1 LORADS_SEPARABLE=0 python bench.py $B > $O/mvec.json 2> $O/mvec.err || exit 1
LORADS_FORCE_DIST=1 python bench.py $B --workload blk16x4000 --times-log-rank 2.0 > $O/sep_cfg4.json 2> $O/sep_cfg4.err || exit 1
LORADS_FORCE_DIST=1 LORADS_SEPARABLE=0 python bench.py $B --workload blk16x4000 --times-log-rank 2.0 > $O/mvec_cfg4.json 2> $O/mvec_cfg4.err || exit 1
python - <<'PY'
import json
for n in ("plain", "sep", "mvec", "sep_cfg4", "mvec_cfg4"):
    d = json.loads(open("gpurun_out/r02dist1/%s.json" % n).read().strip().splitlines()[-1])
    print("%-10s %8.1f it/s  ms/step median %.4f  %s" % (n, d["value"], d["ms_per_step_median"], d["config"]["parallelism"]))
PY
grep -h "hook mode" $O/*.err | sort | uniq -c
