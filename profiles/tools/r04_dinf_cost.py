#!/usr/bin/env python3
"""Cost of the dual-infeasibility eigen-solve after a solve: wall time and operator applications of
lorads_hip_dual_infeasibility at the reference's parameters (tol 1e-2, ncv 40, 600 restarts) and at 1e-6."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lorads_amd import instances  # noqa: E402
from tests import common  # noqa: E402

for name, tlr in [("maxcut4000", 3.0), ("rand4000", 3.0), ("maxcut20000", 4.0), ("blk16x4000", 2.0), ("blk4x60", 2.0)]:
    p = "/tmp/lorads_bench_%s.dat-s" % name
    if not os.path.exists(p):
        instances.write_sdpa(instances.NAMED[name](), p)
    s = common.hip_session(p, timesLogRank=tlr, reoptLevel=0)
    t = time.time()
    s.solve()
    ts = time.time() - t
    for tol in (1e-2, 1e-6):
        best = None
        for rep in range(3):
            t = time.time()
            v, lm, nmv = s.hip_dual_infeasibility(tol=tol)
            dt = time.time() - t
            best = dt if best is None else min(best, dt)
        print("%-12s solve %.3f s | tol %.0e: sum %.6e lam_min %s applications %d  %.4f s" % (name, ts, tol, v, ["%.6e" % x for x in lm[:2]], nmv, best), flush=True)
    s.close()
