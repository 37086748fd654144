"""Phase 1's L-BFGS history update + next direction as ONE launch of resident workgroups (lorads_amd/csrc/hip/lbfgs_team.inc) against
the launch-by-launch form of the same library (LORADS_LBFGS_TEAM=0: k_his_two_dot, 2 nn + 1 k_lbfgs_stage, k_use_grad_p).  Reference:
setlbfgsHisTwo, LBFGSDirection, LBFGSDirectionUseGrad (src_semi/lorads_alg/lorads_alm.c:230-391,469-489,540-560)."""
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


def _session(path, team, **kw):
    """team: True (the default: the one-launch direction and the shared passes behind it), "direction" (the one launch, then the
    launch-by-launch tail: LORADS_ALM_FUSED_TAIL=0), False (LORADS_LBFGS_TEAM=0: everything launch by launch)"""
    os.environ["LORADS_LBFGS_TEAM"] = "1" if team else "0"
    if team == "direction":
        os.environ["LORADS_ALM_FUSED_TAIL"] = "0"
    if team == "nofold":   # (the shared passes, but the constraints' bookkeeping as a pass of its own: LORADS_ALM_FOLD_CV=0)
        os.environ["LORADS_ALM_FOLD_CV"] = "0"
    if team == "sval":     # (the entries' coefficients by k_sval instead of by k_alm_update: LORADS_ALM_SVAL_DIRECT=0)
        os.environ["LORADS_ALM_SVAL_DIRECT"] = "0"
    try:
        return common.hip_session(path, **kw)
    finally:
        os.environ.pop("LORADS_LBFGS_TEAM", None)
        os.environ.pop("LORADS_ALM_FUSED_TAIL", None)
        os.environ.pop("LORADS_ALM_FOLD_CV", None)
        os.environ.pop("LORADS_ALM_SVAL_DIRECT", None)


def _steps(path, team, iters, rho=0.7, **kw):
    """`iters` inner iterations through alm_front / alm_step from the starting point, the line search's scalar code on the host"""
    s = _session(path, team, **kw)
    try:
        be = s.be
        be.init_constr(host.PAIR_RR)
        be.alm_cal_grad(rho)
        rec = []
        front = be.alm_front(rho, 0)
        for it in range(iters):
            p1, p2, coef = front
            tau, _ = common.linesearch_tau(coef)
            lag, err1, np1, np2, ncoef = be.alm_step(rho, tau, it + 1)
            front = (np1, np2, ncoef)
            rec.append([p1, p2, *coef, tau, lag, err1])
        st = s.hip_lbfgs_team_stats()
        mats = [be.get_mat(w, k) for w in (host.MAT_R, host.MAT_U, host.MAT_GRAD) for k in range(s.nblk)]
        return rec, mats, st
    finally:
        s.close()


def _close(a, b, rtol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) <= rtol * max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("name,iters", [("rand120", 8), ("maxcut100", 8), ("theta30", 6), ("matcomp60", 8)])
def test_one_launch_direction_equals_the_stage_by_stage_form(built, name, iters):
    """the same inner iterations from the same start: every scalar the host sees (p1, p2, the line search's coefficients, tau,
    ||Grad||^2, the residual) and R, the direction D, the gradient afterwards -- equal to the rounding of five dots summed in
    another order.  Iteration 0 -> 1 uses one history pair, later ones two."""
    path = common.instance_path(name)
    (ra, ma, sa), (rb, mb, sb), (rc, mc, sc) = _steps(path, True, iters), _steps(path, False, iters), _steps(path, "direction", iters)
    rd, md, sd = _steps(path, "nofold", iters)
    re_, me, se = _steps(path, "sval", iters)
    assert sb["launches"] == 0, sb
    if sa["available"] == 0:
        pytest.skip("%s: not a context the one-launch form applies to (%s)" % (name, sa))
    assert sa["launches"] == iters and sc["launches"] == iters and sd["launches"] == iters, (sa, sc, sd)
    # (the coefficients k_alm_update writes are the ones k_sval would: the same numbers bit for bit)
    assert ra == re_ and all(np.array_equal(x, y) for x, y in zip(ma, me)), name
    for r2, m2 in ((ra, ma), (rc, mc), (rd, md)):
        for i, (x, y) in enumerate(zip(r2, rb)):
            assert _close(x, y, 1e-9), (name, i, x, y)
        for x, y in zip(m2, mb):
            assert _close(x, y, 1e-9), name
    print(name, "ok:", sa)


@pytest.mark.parametrize("hist", [1, 3])
def test_other_history_lengths(built, hist):
    """lbfgsListLength 1 (the one launch with a single pair: the older node is the new one and is never read) and 3 (above what the
    one launch holds: the stage-by-stage form runs, the switch changes nothing) -- 8 inner iterations on rand120, both settings"""
    out = []
    for team in (True, False):
        os.environ["LORADS_LBFGS_TEAM"] = "1" if team else "0"
        try:
            s = host.Session.open(common.instance_path("rand120"))
            s.set_params(verbose=0, lbfgsListLength=hist)
            s.prepare(1, 0, separable=False)
            s.attach_hip(lbfgs_len=hist)
        finally:
            os.environ.pop("LORADS_LBFGS_TEAM", None)
        try:
            be = s.be
            rho = 0.7
            be.init_constr(host.PAIR_RR)
            be.alm_cal_grad(rho)
            rec = []
            front = be.alm_front(rho, 0)
            for it in range(8):
                p1, p2, coef = front
                tau, _ = common.linesearch_tau(coef)
                lag, err1, np1, np2, ncoef = be.alm_step(rho, tau, it + 1)
                front = (np1, np2, ncoef)
                rec.append([p1, p2, *coef, tau, lag, err1])
            out.append((rec, be.get_mat(host.MAT_R, 0), s.hip_lbfgs_team_stats()))
        finally:
            s.close()
    (ra, Ra, sa), (rb, Rb, sb) = out
    assert sb["launches"] == 0 and sa["launches"] == (8 if hist <= 2 else 0), (sa, sb)
    for i, (x, y) in enumerate(zip(ra, rb)):
        assert _close(x, y, 1e-9), (hist, i, x, y)
    assert _close(Ra, Rb, 1e-9)


@pytest.mark.parametrize("name,params", [("maxcut100", dict(reoptLevel=0)), ("rand120", dict(reoptLevel=1, phase1Tol=1e-2)),
                                         ("theta30", dict(reoptLevel=1, phase1Tol=1e-2)), ("matcomp60", dict(reoptLevel=0, phase1Tol=1e-2))])
def test_whole_solves_agree(built, name, params):
    """whole solves with the one-launch direction inside phase 1 against the stage-by-stage form: the same objective; the same
    inner iteration count wherever the instance is not chaotic in the last bits of its dots (all of these)"""
    res = []
    for team in (True, False):
        s = _session(common.instance_path(name), team, **params)
        try:
            s.solve()
            res.append((s.results(), s.hip_lbfgs_team_stats()))
        finally:
            s.close()
    (a, sa), (b, sb) = res
    assert sb["launches"] == 0
    print(name, "inner iterations", a["alm_inner"], b["alm_inner"], "ADMM", a["admm_iter"], b["admm_iter"], "launches", sa["launches"])
    assert abs(a["pObj"] - b["pObj"]) <= 1e-6 * (1 + abs(b["pObj"]))
    assert abs(a["alm_inner"] - b["alm_inner"]) <= max(3, 0.05 * b["alm_inner"]), (a["alm_inner"], b["alm_inner"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,tlr", [("rand20000", 4.0), ("matcomp50000", 2.0), ("maxcut20000", 4.0)])
def test_one_launch_direction_fullsize(built, name, tlr):
    """BASELINE configs 3b, 5 and 3a at full size: 6 inner iterations, both forms (2 ... 6 pairs of doubles per thread and vector)"""
    path = _gen(name)
    (ra, ma, sa), (rb, mb, sb) = _steps(path, True, 6, timesLogRank=tlr), _steps(path, False, 6, timesLogRank=tlr)
    assert sa["available"] == 1 and sa["launches"] == 6 and sb["launches"] == 0, (sa, sb)
    for i, (x, y) in enumerate(zip(ra, rb)):
        assert _close(x, y, 1e-9), (name, i, x, y)
    for x, y in zip(ma, mb):
        assert _close(x, y, 1e-9), name
    print(name, "ok:", sa)
