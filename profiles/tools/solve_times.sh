# whole solves with default parameters on mid-size generated instances (the reference's times for the same files are in
# DESIGN.md section 7); run through gpurun from the repo root
set -e
python - <<'PY'
import sys
sys.path.insert(0, '.')
from lorads_amd import instances
instances.write_sdpa(instances.blockdiag_maxcut(8, 1000, 6000, 4000), '/tmp/blk8x1000.dat-s')
instances.write_sdpa(instances.matcomp(2000, 2000, 16000, 5, 777), '/tmp/matcomp4000.dat-s')
instances.write_sdpa(instances.sdp_lp(2000, 8000, 300, 4100), '/tmp/sdplp2000.dat-s')
instances.write_sdpa(instances.NAMED["maxcut4000"](), '/tmp/maxcut4000.dat-s')
instances.write_sdpa(instances.NAMED["rand4000"](), '/tmp/rand4000.dat-s')
PY
for f in blk8x1000 matcomp4000 sdplp2000; do
  echo "== $f"; lorads_amd/lib/lorads /tmp/$f.dat-s --reoptLevel 1 | grep -E "Primal Objective|Dual Objective|Constraint Violation\(1\)|phase 1:"
done
for f in maxcut4000 rand4000; do
  echo "== $f"; lorads_amd/lib/lorads /tmp/$f.dat-s --timesLogRank 3.0 --reoptLevel 1 | grep -E "Primal Objective|Dual Objective|Constraint Violation\(1\)|phase 1:"
done
