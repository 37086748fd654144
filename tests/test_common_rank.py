"""The cones of a block-separable context share the largest DEVICE rank (lorads_amd/csrc/hip/build.inc: common_rank; zero columns beyond a
cone's own rank, which the caller never sees) so that cones of unequal rank run as one block-diagonal cone -- the lockstep ADMM sweep, the
single-cone forms of phase 1 -- instead of cone by cone.  Against per-cone ranks (LORADS_COMMON_RANK=0).  Reference: the ranks are per cone
(src_semi/data/lorads_solver.c:290-319, AUG_RANK :806-906); what is computed on the extra columns is exact zeros."""
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


def _session(path, on, **kw):
    os.environ["LORADS_COMMON_RANK"] = "1" if on else "0"
    try:
        return common.hip_session(path, **kw)
    finally:
        os.environ.pop("LORADS_COMMON_RANK", None)


@pytest.mark.parametrize("name,params", [("blkmix5", dict(reoptLevel=0, phase1Tol=1e-2)), ("blkmix5", dict(reoptLevel=1, phase1Tol=1e-2, timesLogRank=3.5))])
def test_whole_solves_with_a_common_device_rank(built, name, params):
    """whole solves (rank growth included where the rule asks for it): the same iteration counts, the same objective, factors of the
    cones' OWN shapes equal to rounding"""
    path = _gen(name) if name == "blkmix5" else common.instance_path(name)
    out = []
    for on in (True, False):
        s = _session(path, on, **params)
        try:
            s.solve()
            r = s.results()
            shapes = [s.block_shape(k) for k in range(s.nblk)]
            mats = [s.be.get_mat(host.MAT_R, k) for k in range(s.nblk)]
            out.append((r, shapes, mats))
        finally:
            s.close()
    (a, sa, ma), (b, sb, mb) = out
    assert sa == sb, (sa, sb)                      # (the caller's ranks: untouched)
    assert len(set(r for _, r in sa)) > 1, sa      # (and really unequal)
    for k in ("alm_outer", "alm_inner", "admm_iter", "cg_iter"):
        assert a[k] == b[k], (name, k, a[k], b[k])
    assert abs(a["pObj"] - b["pObj"]) <= 1e-9 * (1 + abs(b["pObj"]))
    for x, y in zip(ma, mb):
        assert x.shape == y.shape and np.max(np.abs(x - y)) <= 1e-7 * max(np.max(np.abs(y)), 1e-300)
    print(name, "ok: ranks", [r for _, r in sa], "inner", a["alm_inner"], "ADMM", a["admm_iter"], "CG", a["cg_iter"])


def test_cones_of_different_kinds_at_unequal_ranks(built):
    """mix4 (a Max-Cut cone, a random-constraint cone, a matrix-completion cone, ...) grown to ranks 11 / 10 / 12 / 14: 30 ADMM
    iterations from the same factors with and without the common device rank -- the same CG counts, scalars and factors to rounding"""
    out = []
    ranks = [11, 10, 12, 14]
    for on in (True, False):
        s = _session(common.instance_path("mix4"), on, phase1Tol=1e-2)
        try:
            s.alm()                      # (the solver's own phase 1 at the rank rule's ranks: 9 everywhere)
            s.alm_to_admm()
            s.be.resize_rank(ranks)      # (AUG_RANK's growth: deterministic new columns)
            assert [s.block_shape(k)[1] for k in range(s.nblk)] == ranks
            s.be.init_constr(host.PAIR_UV)
            res0 = s.results()
            rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
            log = []
            for it in range(30):
                log.append(s.be.admm_step(rho, 1e-8, 800))
                s.be.update_dual_var(rho)
            out.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], s.be.get_vec(host.VEC_LAMBDA)))
        finally:
            s.close()
    (la, Ua, lama), (lb, Ub, lamb) = out
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x[0] == y[0], (i, x, y)
        for q in (1, 2, 3):
            assert abs(x[q] - y[q]) <= 1e-9 * abs(y[q]) + 1e-13, (i, q, x[q], y[q])
    for x, y in zip(Ua + [lama], Ub + [lamb]):
        assert x.shape == y.shape and np.max(np.abs(x - y)) <= 1e-8 * max(np.max(np.abs(y)), 1e-300)


@pytest.mark.timeout(900)
def test_unequal_general_cones_fullsize(built):
    """eight cones of the headline's kind, n = 1500 ... 3250 (ranks 15 ... 17): phase 1 + 20 ADMM iterations both ways -- equal inner
    iteration and CG counts to a percent (two separate runs of a thousand L-BFGS iterations), the same objective; with the common rank the context runs on the merged view (launch counts)"""
    path = _gen("randblk8var")
    out = []
    for on in (True, False):
        s = _session(path, on, phase1Tol=1e-2)
        try:
            n0 = s.hip_launch_count()
            s.alm()
            inner = s.results()["alm_inner"]
            n_p1 = s.hip_launch_count() - n0
            s.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
            res0 = s.results()
            rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
            log = []
            n0 = s.hip_launch_count()
            for it in range(20):
                log.append(s.be.admm_step(rho, 1e-8, 800))
                s.be.update_dual_var(rho)
            out.append((inner, log, n_p1 / inner, (s.hip_launch_count() - n0) / 20.0))
        finally:
            s.close()
    (ia, la, pa, qa), (ib, lb, pb, qb) = out
    # (a thousand L-BFGS iterations amplify the last bit of their dots: 1060 against 1062 -- the exact comparisons are the small tests above)
    assert abs(ia - ib) <= max(3, 0.01 * ib), (ia, ib)
    ca, cb = sum(x[0] for x in la), sum(x[0] for x in lb)
    assert abs(ca - cb) <= 0.05 * cb, (ca, cb)
    assert abs(la[-1][1] - lb[-1][1]) <= 1e-4 * abs(lb[-1][1]), (la[-1], lb[-1])
    assert pa < 0.25 * pb and qa < 0.5 * qb, (pa, pb, qa, qb)
    print("randblk8var ok: launches per inner iteration %.1f (%.1f cone by cone), per ADMM iteration %.1f (%.1f)" % (pa, pb, qa, qb))
