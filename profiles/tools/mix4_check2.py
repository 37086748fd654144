import sys, os
sys.path.insert(0, '.')
from tests import common
from lorads_amd import host
import numpy as np
for env in ({}, {"LORADS_NO_MERGE": "1"}, {"LORADS_NO_BATCH": "1"}):
    for k in ("LORADS_NO_MERGE", "LORADS_NO_BATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with common.hip_session(common.instance_path("mix4"), reoptLevel=0) as s:
        s.solve(); r = s.results(); lam = s.be.get_vec(host.VEC_LAMBDA)
        print(env, "p=%.8f d=%.8f vio=%.2e alm=%d admm=%d |lam|=%.10f" % (r["pObj"], r["dObj"], r["constrVio1"], r["alm_inner"], r["admm_iter"], np.linalg.norm(lam)), flush=True)
for k in ("LORADS_NO_MERGE", "LORADS_NO_BATCH"):
    os.environ.pop(k, None)
with common.oracle_session(common.instance_path("mix4"), reoptLevel=0) as s:
    s.solve(); r = s.results(); lam = s.be.get_vec(host.VEC_LAMBDA)
    print("oracle", "p=%.8f d=%.8f vio=%.2e alm=%d admm=%d |lam|=%.10f" % (r["pObj"], r["dObj"], r["constrVio1"], r["alm_inner"], r["admm_iter"], np.linalg.norm(lam)))
