#!/usr/bin/env python3
"""profiles/<round>_stamp.json: which HIP sources (and, when run in the build container, which commit) the round's committed
rocprofv3 / PMC summaries were collected from.  bench.py replays a committed profile into its JSON line only if the stamp's
source hash equals the hash of the sources it is running (a profile of older kernels is not this run's evidence).
usage: stamp.py <round, e.g. r02>   (run on the GPU box right after collecting; run again in the container to add the commit)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rnd = args[0]
out = os.path.join(ROOT, "profiles" if len(args) < 2 else args[1], "%s_stamp.json" % rnd)
st = {}
if os.path.exists(out):
    with open(out) as fh:
        st = json.load(fh)
h = bench.hip_source_hash()
if st.get("hip_source_sha256") not in (None, h) and "--keep" in sys.argv:
    sys.exit("sources changed since the stamp was written")
st["hip_source_sha256"] = h
try:
    st["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
except Exception:  # noqa: BLE001  (no repository on the GPU box)
    st.setdefault("git_head", None)
with open(out, "w") as fh:
    json.dump(st, fh, indent=1)
print(out, st)
