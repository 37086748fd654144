import sys
sys.path.insert(0, '.')
from tests import common
for params in [dict(dyrankLevel=0, reoptLevel=1), dict(dyrankLevel=0, reoptLevel=2), dict(dyrankLevel=0, reoptLevel=0),
               dict(highAccMode=1, reoptLevel=1), dict(lbfgsListLength=4, reoptLevel=1), dict(reoptLevel=2)]:
    out = []
    for mk in (common.hip_session, common.oracle_session):
        with mk(common.instance_path("mix4"), **params) as s:
            s.solve()
            r = s.results()
            out.append("%s p=%.6f d=%.6f vio=%.1e dinf=%.1e st=%d alm=%d admm=%d" % (mk.__name__[:3], r["pObj"], r["dObj"], r["constrVio1"],
                       r["dual_infeas_l1"], r["status"], r["alm_inner"], r["admm_iter"]))
    print(params, " | ".join(out), flush=True)
