import os, sys, time
sys.path.insert(0, os.getcwd())
from lorads_amd import host, instances
for name, tlr in [("maxcut800", 2.0), ("maxcut800", 2.0), ("maxcut4000", 2.0), ("maxcut800", 2.0)]:
    path = "/tmp/lorads_bench_%s.dat-s" % name
    if not os.path.exists(path):
        instances.write_sdpa(instances.NAMED[name](), path)
    s = host.Session.open(path)
    s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
    s.prepare(1, 0, separable=False)
    s.attach_hip()
    s.hip_sync()
    n0 = s.hip_launch_count()
    t = time.perf_counter(); s.alm(); s.hip_sync(); t = time.perf_counter() - t
    res = s.results()
    print(name, "inner", res["alm_inner"], "outer", res["alm_outer"], "%.1f ms" % (t * 1e3), "%.1f us/inner" % (t * 1e6 / res["alm_inner"]), "launches/inner %.1f" % ((s.hip_launch_count() - n0) / res["alm_inner"]))
    s.close()
