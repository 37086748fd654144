import sys, time
sys.path.insert(0, '.')
from tests import common
from lorads_amd import instances
import os
for name, tlr in [("maxcut4000", 3.0), ("rand4000", 3.0), ("blk4x60", 2.0)]:
    p = "/tmp/%s.dat-s" % name
    if not os.path.exists(p):
        instances.write_sdpa(instances.NAMED[name](), p)
    s = common.hip_session(p, timesLogRank=tlr, reoptLevel=0)
    s.solve()
    for tol in (1e-2, 1e-6):
        t = time.time(); v, lm, nmv = s.hip_dual_infeasibility(tol=tol); dt = time.time() - t
        print(name, "tol", tol, "sum", v, "lam_min", lm[:2], "matvecs", nmv, "%.4f s" % dt, flush=True)
    s.close()
