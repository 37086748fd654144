"""How the hot path scales with the cone's size: the two headline families (random sparse A_i as cfg3b, Max-Cut as cfg3a) at rank 40
from n = 2500 to n = 80000 -- ADMM iterations / s, CG iterations / s, the live CG operator back to back (lorads_hip_time_operator)
against the algorithmic bytes of SURVEY 8d.  Small cones are bound by the ~1.7 us a kernel boundary costs (10 launches per
iteration), large ones by the rate at which the fabric delivers gathered rows.  usage: size_sweep.py [steps]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lorads_amd import host, instances  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
print("%-10s %7s %4s %9s | %9s %10s %7s | %8s %9s %6s | %s" % ("family", "n", "r", "m", "ADMM it/s", "CG it/s", "ms/it", "op us", "alg MB", "% HBM", "operator"))
for fam in ("rand", "maxcut"):
    for n in (2500, 5000, 10000, 20000, 40000, 80000):
        name = "%s%d" % (fam, n)
        if name not in instances.NAMED:
            instances.NAMED[name] = (lambda n=n: instances.randsparse(n, n // 4, n + 1, c_edges=6 * n)) if fam == "rand" else \
                                    (lambda n=n: instances.maxcut(n, 6 * n, n))
        path = bench.build_instance(name, "/tmp/lorads_sweep_%s.dat-s" % name)
        tlr = 39.5 / math.log(n)          # r = ceil(t log n) = 40
        s = host.Session.open(path)
        s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
        s.prepare(1, 0)
        s.attach_hip()
        s.hip_sync()
        n_l0 = s.hip_launch_count()
        t_p1 = time.perf_counter()
        s.alm()
        s.hip_sync()
        t_p1 = time.perf_counter() - t_p1
        n_l_p1 = s.hip_launch_count() - n_l0
        s.alm_to_admm()
        res = s.results()
        rho = min(res["admm_rho"] if res["admm_rho"] > 0 else res["alm_rho"], 5000.0)
        be = s.be
        be.init_constr(host.PAIR_UV)
        be.cal_obj(host.PAIR_UV)
        err1 = be.update_dimacs(host.PAIR_UV)
        err1, _, _, _ = bench.admm_steps(be, host, rho, err1, 10, s)
        s.hip_sync()
        n_l0 = s.hip_launch_count()
        t0 = time.perf_counter()
        err1, cg, _, _ = bench.admm_steps(be, host, rho, err1, steps, s)
        s.hip_sync()
        el = time.perf_counter() - t0
        n_l = s.hip_launch_count() - n_l0
        op_ms = s.hip_time_operator(200) / 200
        mv, _ = s.hip_algorithmic_bytes(0)
        info = s.block_info(0)
        gs = s.hip_graph_stats()
        inner = max(res["alm_inner"], 1)
        print("%-10s %7d %4d %9d | %9.1f %10.1f %7.4f | %8.2f %9.2f %6.1f | %s | graph replays %d captures %d | launches per ADMM iteration %.1f | "
              "phase 1: %.1f us and %.1f launches per inner iteration (%d)" %
              (fam, n, info["rank"], info["nrow"], steps / el, cg / el, 1e3 * el / steps, 1e3 * op_ms, mv / 1e6, 100 * mv / (op_ms * 1e-3) / 8e12,
               s.hip_operator_kind(0), gs["replayed"], gs["captured"], n_l / steps, 1e6 * t_p1 / inner, n_l_p1 / inner, int(inner)), flush=True)
        s.close()
