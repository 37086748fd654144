import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
from tests import common
from lorads_amd import host, instances
import os
p = "/tmp/rand4000.dat-s"
if not os.path.exists(p):
    instances.write_sdpa(instances.NAMED["rand4000"](), p)
s = common.hip_session(p, timesLogRank=3.0, reoptLevel=1)
s.solve()
lib, ctx = s._hip()
lib.lorads_hip_dual_infeasibility.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
for tol in (1e-2, 1e-2, 1e-3, 1e-1):
    sn, lm, nmv = C.c_double(), (C.c_double * 4)(), C.c_int()
    t0 = time.time()
    rc = lib.lorads_hip_dual_infeasibility(ctx, tol, 40, 600, C.byref(sn), lm, C.byref(nmv))
    print("tol", tol, "rc", rc, "sum_neg", sn.value, "lam_min", lm[0], "matvecs", nmv.value, "time %.3f" % (time.time() - t0))
s.close()
