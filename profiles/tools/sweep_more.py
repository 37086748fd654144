import sys
sys.path.insert(0, '.')
from tests import common
SWEEP = [dict(dyrankLevel=0), dict(dyrankLevel=3), dict(highAccMode=1), dict(lbfgsListLength=4), dict(reoptLevel=2, phase1Tol=1e-2),
         dict(initRho=0.5), dict(timesLogRank=0.5), dict(timesLogRank=0.0), dict(phase2Tol=1e-7), dict(reoptLevel=2, phase1Tol=1e-2, highAccMode=1)]
bad = 0
for name in ["coupled3x70", "densec40", "matcomp60", "maxcut100", "theta30", "sdpslack30", "maxcut800"]:
    for params in SWEEP:
        res = []
        for mk in (common.hip_session, common.oracle_session):
            with mk(common.instance_path(name), **params) as s:
                s.solve(); res.append(s.results())
        h, o = res
        tol = max(2e-4, 10 * o["pdGap"], 10 * h["pdGap"])
        ok = h["constrVio1"] <= 2e-5 and o["constrVio1"] <= 2e-5 and abs(h["pObj"] - o["pObj"]) <= tol * (1 + abs(o["pObj"]))
        if not ok:
            bad += 1
        print("%-12s %-55s %s hip p=%.6f vio=%.1e gap=%.1e dinf=%.1e | ora p=%.6f vio=%.1e gap=%.1e dinf=%.1e" % (name, params, "ok " if ok else "BAD",
              h["pObj"], h["constrVio1"], h["pdGap"], h["dual_infeas_l1"], o["pObj"], o["constrVio1"], o["pdGap"], o["dual_infeas_l1"]), flush=True)
print("bad:", bad)
