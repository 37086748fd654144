#!/usr/bin/env python3
"""Phase times of the one-launch ADMM iteration (persist.inc) from the leader workgroup's 100 MHz clock stamps.
usage (GPU box): python profiles/tools/r04_persist_stamps.py maxcut800:2.0 blk16x4000:2.0 maxcut20000:4.0"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lorads_amd import host, instances  # noqa: E402


def main():
    for spec in sys.argv[1:]:
        name, tlr = spec.split(":")
        path = "/tmp/lorads_bench_%s.dat-s" % name
        if not os.path.exists(path):
            instances.write_sdpa(instances.NAMED[name](), path)
        s = host.Session.open(path)
        s.set_params(verbose=0, timesLogRank=float(tlr), phase1Tol=1e-2, reoptLevel=0)
        s.prepare(1, 0, separable=False)
        s.attach_hip()
        try:
            s.alm()
            s.alm_to_admm()
            res = s.results()
            rho = min(res["admm_rho"] if res["admm_rho"] > 0 else res["alm_rho"], 5000.0)
            be = s.be
            be.init_constr(host.PAIR_UV)
            be.cal_obj(host.PAIR_UV)
            err1 = be.update_dimacs(host.PAIR_UV)
            err1, _, _, _ = s.admm_steps(30, rho, err1)
            s.hip_persist_stamps(True)
            rows = []
            cgs = []
            for _ in range(40):
                t0 = time.perf_counter()
                c, p, d, err1 = be.admm_step(rho, min(err1 * 1e-2, 1e-8), 800)
                wall = time.perf_counter() - t0
                be.update_dual_var(rho)
                st = s.hip_persist_stamps(True)
                if st[0] == 0:
                    continue
                k = max(i for i in range(16) if st[i])
                rows.append([(st[i + 1] - st[i]) * 0.01 for i in range(min(k, 6))] + [wall * 1e6])
                cgs.append(c)
            a = np.array([r for r in rows if len(r) == len(rows[0])])
            med = np.median(a, axis=0)
            names = ["U front", "U solve (CG)", "V front", "V solve (CG)", "evaluation", "report"][:a.shape[1] - 1] + ["host wall of the step"]
            print("%s (persist %s): CG its per step %.2f" % (name, s.hip_persist_stats(), float(np.mean(cgs))))
            for nm, v in zip(names, med):
                print("    %-24s %8.2f us" % (nm, v))
            print("    %-24s %8.2f us" % ("sum in kernel", float(np.sum(med[:-1]))))
        finally:
            s.close()


if __name__ == "__main__":
    main()
