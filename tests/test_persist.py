"""The one-launch ADMM iteration of Max-Cut-type cones (lorads_amd/csrc/hip/persist.inc: teams of resident workgroups, factors in
registers, in-kernel all-reduces) against the launch-by-launch form of the same library (LORADS_PERSIST=0) and, at full size,
against the compiled reference.  Reference loops: lorads_alg/lorads_alg_common.c:187-215, linalg/lorads_cgs.c:174-234."""
import os
import sys

import numpy as np
import pytest

from lorads_amd import host
from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    import __graft_entry__
    __graft_entry__.build()


def _gen(name):
    path = os.path.join("/tmp", "lorads_test_%s.dat-s" % name)
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
        import gen_instances
        gen_instances.write_sdpa(gen_instances.NAMED[name](), path)
    return path


def _session(path, persist, **kw):
    os.environ["LORADS_PERSIST"] = "1" if persist else "0"
    try:
        return common.hip_session(path, **kw)
    finally:
        os.environ.pop("LORADS_PERSIST", None)


def _run(path, persist, iters, kw, mixed=True):
    s = _session(path, persist, **kw)
    try:
        s.alm()
        s.alm_to_admm()
        s.be.init_constr(host.PAIR_UV)
        res0 = s.results()
        rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
        tols = [1e-8, 1e-8, 1e-8, 1e-8, 1e-3, 1e-8, 1e-8, 1e-12, 1e-8, 1e-8]
        log = []
        for it in range(iters):
            if mixed and it % 7 == 6:   # the slot-by-slot entries: sweep without an evaluation, then the evaluation's three calls
                c = s.be.admm_update_var(rho, tols[it % len(tols)], 800)
                p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
            else:
                c, p, d, e = s.be.admm_step(rho, tols[it % len(tols)], 800 if (it != 33 or not mixed) else 3)
            s.be.update_dual_var(rho)
            if it % 10 == 9:
                rho *= 1.2
            log.append((c, p, d, e))
        st = s.hip_persist_stats()
        U = [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)]
        V = [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]
        R = [s.be.get_mat(host.MAT_R, k) for k in range(s.nblk)]
        return log, U, V, R, s.be.get_vec(host.VEC_LAMBDA), s.be.get_vec(host.VEC_CONSTR_SUM), st
    finally:
        s.close()


def _compare(a, b, iters, tag):
    (la, Ua, Va, Ra, lama, csa, sta), (lb, Ub, Vb, Rb, lamb, csb, stb) = a, b
    assert sta["available"] == 1 and sta["iterations"] >= iters, sta
    assert stb["available"] == 0 and stb["iterations"] == 0, stb
    worst = 0.0
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x[0] == y[0], (tag, "CG iterations differ at ADMM iteration", i, x, y)   # the same decisions, iteration by iteration
        for q in (1, 2):
            worst = max(worst, abs(x[q] - y[q]) / max(abs(y[q]), 1e-30))
        # (the residual norm b - A(R R^T) is a difference of numbers of size one: its own rounding is 1e-16 absolute, whatever its size)
        assert abs(x[3] - y[3]) <= 1e-9 * abs(y[3]) + 1e-14, (tag, i, x[3], y[3])
    assert worst <= 1e-9, (tag, worst)
    for x, y in zip(Ua + Va + Ra + [lama, csa], Ub + Vb + Rb + [lamb, csb]):
        sc = max(np.max(np.abs(y)), 1e-300)
        assert np.max(np.abs(x - y)) <= 1e-9 * sc, (tag, np.max(np.abs(x - y)) / sc)
    print(tag, "ok: %d ADMM iterations, %d CG iterations, worst scalar difference %.1e, workgroups %d, rows %d" %
          (len(la), int(sum(x[0] for x in la)), worst, sta["workgroups"], sta["rows"]))


@pytest.mark.parametrize("name,tlr", [("maxcut100", None), ("maxcut800", None), ("blk4x60", None), ("blkmix5", None), ("blkmix5", 3.5), ("maxcut100", 7.0)])
def test_one_launch_iteration_equals_launch_by_launch(built, name, tlr):
    """70 ADMM iterations from the solver's own phase 1 with changing penalty, tolerance and iteration limit, the fused step and the
    slot-by-slot entries mixed, a pending dual update on board of every sweep: the same CG iteration counts and the same numbers
    (sums in another order: 1e-9) as the launch-by-launch form.  blkmix5: five cones of different size AND rank -- no lockstep,
    no equal-rank condition (ranks 9 ... 12; at timesLogRank 3.5: 16 ... 20, i.e. one and two column steps in one launch).
    maxcut100 at timesLogRank 7: rank 33 -> three column steps."""
    path = _gen(name) if name in ("blkmix5",) else common.instance_path(name)
    kw = dict(phase1Tol=1e-2)
    if tlr:
        kw["timesLogRank"] = tlr
    res = [_run(path, p, 70, kw) for p in (True, False)]
    _compare(res[0], res[1], 60, name)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,tlr,iters", [("blk16x4000", None, 24), ("maxcut20000", 4.0, 24), ("blk16var", None, 16)])
def test_one_launch_iteration_equals_launch_by_launch_fullsize(built, name, tlr, iters):
    """BASELINE cfg4 (16 cones n = 4000), cfg3a (Max-Cut n = 20000, r = 40) and a cfg4-shaped problem with unequal cones
    (n_k = 2000 ... 5750, ranks 16 ... 18) at full size."""
    path = _gen(name)
    kw = dict(phase1Tol=1e-2)
    if tlr:
        kw["timesLogRank"] = tlr
    res = [_run(path, p, iters, kw, mixed=False) for p in (True, False)]
    _compare(res[0], res[1], iters, name)


def test_carried_objective_gather_changes_nothing_but_the_objectives_rounding(built, monkeypatch):
    """LORADS_PERSIST_CARRY (default on): the evaluation gathers V instead of R, forms (C R) = ((C U) + (C V)) / 2 from the V front's
    (C U), and leaves (C V) for the next launch's U front (two gathers per iteration instead of three).  Against every front
    gathering for itself: the factors, the multipliers and the CG counts are the SAME numbers bit for bit (the U front's operand is
    the sum it would have formed), the objective <C, R R^T> differs by the rounding of one addition.  Mixed with sweeps that do not
    evaluate, slot-by-slot entries, a changed iteration limit and set_mat in between (all of which must drop what was carried)."""
    out = []
    for carry in ("1", "0"):
        monkeypatch.setenv("LORADS_PERSIST_CARRY", carry)
        s = _session(common.instance_path("blk4x60"), True, phase1Tol=1e-2)
        try:
            s.alm()
            s.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
            rho, log = 1.3, []
            for it in range(40):
                if it % 6 == 5:
                    c = s.be.admm_update_var(rho, 1e-8, 800)
                    p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
                else:
                    c, p, d, e = s.be.admm_step(rho, 1e-8 if it % 4 else 1e-3, 800 if it != 17 else 2)
                s.be.update_dual_var(rho)
                if it == 22:   # the caller rewrites a factor: what was carried belongs to the old V
                    V0 = s.be.get_mat(host.MAT_V, 0)
                    s.be.set_mat(host.MAT_V, 0, 0.5 * V0)
                    s.be.init_constr(host.PAIR_UV)
                log.append((c, p, d, e))
            out.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)],
                        s.be.get_vec(host.VEC_LAMBDA)))
        finally:
            s.close()
    (la, Ua, Va, lama), (lb, Ub, Vb, lamb) = out
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x[0] == y[0] and x[2] == y[2] and x[3] == y[3], (i, x, y)
        assert abs(x[1] - y[1]) <= 1e-13 * abs(y[1]), (i, x[1], y[1])
    for x, y in zip(Ua + Va + [lama], Ub + Vb + [lamb]):
        assert np.array_equal(x, y)


def test_rank_growth_rebuilds_the_teams(built):
    """AUG_RANK between two sweeps (data/lorads_solver.c:806-906): the plan follows the new ranks"""
    s = _session(common.instance_path("blk4x60"), True, phase1Tol=1e-2)
    try:
        s.alm()
        s.alm_to_admm()
        s.be.init_constr(host.PAIR_UV)
        a = s.be.admm_step(1.0, 1e-8, 800)
        st0 = s.hip_persist_stats()
        shapes = [s.block_shape(k) for k in range(s.nblk)]
        s.be.resize_rank([min(n, r + 5 + k) for k, (n, r) in enumerate(shapes)])
        s.be.init_constr(host.PAIR_UV)
        b = s.be.admm_step(1.0, 1e-8, 800)
        st1 = s.hip_persist_stats()
        assert st0["available"] == 1 and st1["available"] == 1 and st1["iterations"] == st0["iterations"] + 1, (st0, st1)
        assert np.isfinite(a[1]) and np.isfinite(b[1])
    finally:
        s.close()
    # ... and gives the launch-by-launch numbers at the new ranks
    out = []
    for persist in (True, False):
        s = _session(common.instance_path("blk4x60"), persist, phase1Tol=1e-2)
        try:
            s.alm()
            s.alm_to_admm()
            shapes = [s.block_shape(k) for k in range(s.nblk)]
            s.be.resize_rank([min(n, r + 5 + k) for k, (n, r) in enumerate(shapes)])
            s.be.init_constr(host.PAIR_UV)
            log = [s.be.admm_step(1.0, 1e-8, 800) for _ in range(10)]
            out.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)]))
        finally:
            s.close()
    (la, Ua), (lb, Ub) = out
    for x, y in zip(la, lb):
        assert x[0] == y[0] and abs(x[1] - y[1]) <= 1e-10 * abs(y[1]), (x, y)
    for x, y in zip(Ua, Ub):
        assert np.max(np.abs(x - y)) <= 1e-9 * np.max(np.abs(y))


def test_whole_solves_take_the_reference_iteration_counts(built):
    """whole solves through the host loop (csrc/host/solver.c) with the one-launch iteration inside: the iteration counts and
    objectives of the compiled reference's own runs (tests/golden/solve.json) on the Max-Cut instances"""
    ran = 0
    for e in common.golden_solves():
        if e["instance"] not in ("maxcut100", "blk4x60") or e["reopt_rounds"] != 0:
            continue
        flags = e["flags"]
        params = {flags[i][2:]: float(flags[i + 1]) if "." in flags[i + 1] or "e" in flags[i + 1] else int(flags[i + 1])
                  for i in range(0, len(flags), 2)}
        s = _session(common.instance_path(e["instance"]), True, **params)
        try:
            r = s.solve()
            st = s.hip_persist_stats()
        finally:
            s.close()
        assert st["iterations"] >= int(e["admm_iter"]) > 0 or int(e["admm_iter"]) == 0, (e["instance"], flags, st, e["admm_iter"])
        assert int(r["admm_iter"]) == int(e["admm_iter"]) and int(r["cg_iter"]) == int(e["admm_cg_iter"]), (e["instance"], flags, r["admm_iter"],
                                                                                                            e["admm_iter"], r["cg_iter"], e["admm_cg_iter"])
        assert abs(r["pObj"] - e["pObj"]) <= 1e-9 * (1 + abs(e["pObj"])), (r["pObj"], e["pObj"])
        ran += 1
    assert ran >= 2
