import sys, time
sys.path.insert(0, '.')
from lorads_amd import host, instances
import os
for name, tlr in [("matcomp50000", 5.5), ("rand20000", 4.0), ("maxcut20000", 4.0), ("blk16x4000", 2.0)]:
    p = "/tmp/%s.dat-s" % name
    if not os.path.exists(p):
        instances.write_sdpa(instances.NAMED[name](), p)
    t0 = time.time(); s = host.Session.open(p); t1 = time.time()
    s.set_params(verbose=0, timesLogRank=tlr); s.prepare(1, 0); t2 = time.time()
    s.attach_hip(); t3 = time.time()
    print("%-14s read %.3f s  rank+start %.3f s  hip create+upload %.3f s" % (name, t1 - t0, t2 - t1, t3 - t2), flush=True)
    s.close()
