"""Parity tests proper: the HIP C-ABI path against (a) golden vectors produced by the compiled
REFERENCE and (b) the CPU oracle on larger seeded inputs.  Run on a real MI355X (-m gpu)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from lorads_amd import host, instances
from tests import common

pytestmark = pytest.mark.gpu

TRACE_NAMES = ["maxcut100", "theta30", "rand120", "blk4x60", "coupled3x70", "densec40", "densea40", "matcomp60", "mix4", "sdplp40", "sdpslack30", "coupledlp"]


@pytest.mark.parametrize("name", TRACE_NAMES)
def test_trace_vs_reference_golden(built, name):
    """every lorads_func slot, inputs and expected outputs from the reference itself (tolerance:
    FP64 with a different summation order -> 1e-9 relative to the vector's scale; CG outputs 1e-5)"""
    g = common.golden_trace(name)
    s = common.hip_session(common.instance_path(name))
    try:
        assert s.be.name == "hip-gfx950"
        log = common.replay_trace(s, g, rtol=1e-9, resync=True)
        w = common.trace_worst(log)
        print(name, "worst rel-to-scale errors", w)
        # what is achieved, not just what the slot-by-slot bounds allow (a 1e4x regression must not pass): phase 1 has 9e-13 at worst
        # (theta30, the dense branch), the ADMM part 3e-14
        assert w["phase1"] <= 1e-10 and w["factors"] <= 1e-11 and w["vectors"] <= 1e-11 and w["objectives"] <= 1e-12, w
    finally:
        s.close()


@pytest.mark.parametrize("form", ["m-vector, one collective per dot", "m-vector, Gram-form direction", "separable, scalars only"])
@pytest.mark.parametrize("name", ["rand120", "blk4x60", "mix4", "coupled3x70", "sdplp40", "densea40"])
def test_sharded_forms_replay_the_reference_trace(built, monkeypatch, name, form):
    """The code paths of sharded cones against the reference's own numbers: one rank with an all-reduce hook (the sum over one rank
    is the identity), every lorads_func slot replayed from the golden trace -- in the m-vector form (constrValSum, q1, q2 through the
    hook) with the L-BFGS direction by the sequential recursion (one collective per dot) and in Gram form (one collective for the
    direction), and in the scalars-only form of separable shards (lorads_hip_set_separable)."""
    monkeypatch.setenv("LORADS_LBFGS_GRAM", "0" if "per dot" in form else "1")
    g = common.golden_trace(name)
    s = common.hip_session(common.instance_path(name), separable=form.startswith("separable"))
    calls = []
    try:
        assert s.separable == form.startswith("separable")
        s.set_allreduce(lambda ptr, count, on_device: calls.append((count, on_device)))
        log = common.replay_trace(s, g, rtol=1e-9, resync=True)
        w = common.trace_worst(log)
        sizes = sorted({c for c, _ in calls})
        print(name, form, "worst rel-to-scale errors", w, "collectives", len(calls), "sizes", sizes)
        assert w["phase1"] <= 1e-9 and w["factors"] <= 1e-10 and w["vectors"] <= 1e-10 and w["objectives"] <= 1e-11, w
        assert calls, "the hook was never called: not the sharded path"
        if form.startswith("separable"):
            # scalars only (at most the 15 products of the Gram-form direction): nothing of length m crosses the ranks
            assert max(sizes) <= 66 and not any(c in (s.m, s.m + 2, 2 * s.m + 2) for c in sizes), sizes
        else:
            assert max(sizes) >= s.m
    finally:
        s.close()


def _pair(path, **params):
    return common.hip_session(path, **params), common.oracle_session(path, **params)


def _gen(name):
    path = os.path.join("/tmp", "lorads_test_%s.dat-s" % name)
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
        import gen_instances
        gen_instances.write_sdpa(gen_instances.NAMED[name](), path)
    return path


@pytest.mark.parametrize("name,tlr", [("maxcut800", 2.0), ("maxcut4000", 3.0), ("rand4000", 3.0), ("densec300", 2.0),
                                      ("densec300", 7.0), ("densea300", 2.0), ("densea300", 6.5), ("denseac200", 3.0)])
def test_functions_vs_oracle_midsize(built, name, tlr):
    """same seeded input through both tables, function by function (sizes the oracle finishes in seconds)"""
    path = common.instance_path(name) if name == "maxcut800" else _gen(name)
    hs, os_ = _pair(path, timesLogRank=tlr)
    try:
        if name.startswith("densea"):   # the dense constraint matrices really take the dense GEMM
            assert "k_dense_cx_b(dense A_i)" in hs.hip_operator_kind(0)
        rho = 0.5
        for it in range(3):
            vals = []
            for s in (hs, os_):
                be = s.be
                if it == 0:
                    be.init_constr(host.PAIR_RR)
                lag = be.alm_cal_grad(rho)
                be.lbfgs_direction(it)
                p1, p2 = be.alm_q12p12()
                k = be.alm_linesearch_coeffs(rho, p1, p2)
                vals.append((lag, p1, p2, k, be.get_mat(host.MAT_U, 0), be.get_vec(host.VEC_Q1), be.get_vec(host.VEC_Q2)))
            (la, p1a, p2a, ka, Da, q1a, q2a), (lb, p1b, p2b, kb, Db, q1b, q2b) = vals
            assert np.isclose(la, lb, rtol=1e-10)
            assert np.isclose(p1a, p1b, rtol=1e-9, atol=1e-9 * abs(p2b))
            assert np.isclose(p2a, p2b, rtol=1e-9)
            assert np.allclose(ka, kb, rtol=1e-8, atol=1e-9 * max(abs(x) for x in kb))
            assert np.allclose(Da, Db, rtol=0, atol=1e-9 * np.abs(Db).max())
            assert np.allclose(q1a, q1b, rtol=0, atol=1e-10 * np.abs(q1b).max())
            assert np.allclose(q2a, q2b, rtol=0, atol=1e-10 * np.abs(q2b).max())
            tau, _ = common.linesearch_tau(kb)
            for s in (hs, os_):
                be = s.be
                be.set_y_as_neg_grad()
                be.alm_update_var(tau)
                be.alm_cal_grad(rho)
                be.set_lbfgs_his_two(tau)
            ea, eb = hs.be.update_dimacs(host.PAIR_RR), os_.be.update_dimacs(host.PAIR_RR)
            assert np.isclose(ea, eb, rtol=1e-9)
            # keep the two states identical so that errors do not compound
            hs.be.set_mat(host.MAT_R, 0, os_.be.get_mat(host.MAT_R, 0))
        # one ADMM iteration incl. both CG solves
        for s in (hs, os_):
            s.be.update_dual_var(rho)
            s.be.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
        ia = hs.be.admm_update_var(2.0, 1e-8, 800)
        ib = os_.be.admm_update_var(2.0, 1e-8, 800)
        assert abs(ia - ib) <= max(2, 0.03 * ib), (ia, ib)
        Ua, Ub = hs.be.get_mat(host.MAT_U, 0), os_.be.get_mat(host.MAT_U, 0)
        Va, Vb = hs.be.get_mat(host.MAT_V, 0), os_.be.get_mat(host.MAT_V, 0)
        assert np.allclose(Ua, Ub, rtol=0, atol=2e-6 * np.abs(Ub).max())
        assert np.allclose(Va, Vb, rtol=0, atol=2e-6 * np.abs(Vb).max())
        pa, pb = hs.be.cal_obj(host.PAIR_UV), os_.be.cal_obj(host.PAIR_UV)
        assert np.isclose(pa, pb, rtol=1e-6)
        ea, eb = hs.be.update_dimacs(host.PAIR_UV), os_.be.update_dimacs(host.PAIR_UV)
        assert np.isclose(ea, eb, rtol=1e-4, atol=1e-9)
    finally:
        hs.close()
        os_.close()


def _flags_to_params(flags):
    return {flags[i][2:]: float(flags[i + 1]) if "." in flags[i + 1] or "e" in flags[i + 1] else int(flags[i + 1])
            for i in range(0, len(flags), 2)}


def _solve_id(i):
    e = common.golden_solves()[i]
    # (the two dense-branch instances with several stationary points are held to the level the reference itself has converged to,
    # 5e-4; what that bound could hide is excluded by the same-start comparison at 1e-8 in
    # test_fullsize_admm_iterations_from_the_devices_own_state_vs_compiled_reference)
    tag = "-objective to the references own convergence level 5e-4" if e["instance"] in ("matcomp60", "theta50") else ""
    return "%s%s%s" % (e["instance"], "".join(e["flags"]).replace("--", "-"), tag)


@pytest.mark.parametrize("idx", range(len(common.golden_solves())), ids=_solve_id)
def test_whole_solve_vs_reference(built, idx):
    """Whole solves against the reference's own runs of the same command (tests/golden/solve.json), in the north-star's
    wording: converged objectives to 1e-6 relative, DIMACS errors no worse than the reference's, and -- where the
    reference's run is in sparse mode and has no reopt round (a few thousand iterations at most: rounding has not yet
    separated the two trajectories) -- the SAME phase-1 inner / ADMM / CG iteration counts, i.e. every stopping, rho and
    restart decision taken alike.  profiles/tools/solve_parity_report.py prints the numbers behind these assertions."""
    e = common.golden_solves()[idx]
    params = _flags_to_params(e["flags"])
    s = common.hip_session(common.instance_path(e["instance"]), **params)
    try:
        r = s.solve()
    finally:
        s.close()
    ref_gap = abs(e["pObj"] - e["dObj"]) / (1 + abs(e["pObj"]) + abs(e["dObj"]))
    # two instances where the reference's dense branch (dsyr2k / dsymm rounding) sends phase 1 down another path and the
    # rank-constrained problem has more than one point meeting the stopping rule (see tests/test_oracle_vs_reference.py)
    loose = e["instance"] in ("matcomp60", "theta50")
    tol = max(5e-4 if loose else 1e-6, 5 * ref_gap)
    assert abs(r["pObj"] - e["pObj"]) <= tol * (1 + abs(e["pObj"])), (r["pObj"], e["pObj"])
    assert abs(r["dObj"] - e["dObj"]) <= tol * (1 + abs(e["dObj"])), (r["dObj"], e["dObj"])
    p2 = params.get("phase2Tol", 1e-5)
    k = 3.0 if loose else 2.0
    assert r["constrVio1"] <= max(k * e["err_constr_l1"], p2), (r["constrVio1"], e["err_constr_l1"])
    assert r["pdGap"] <= max(k * e["err_pdgap"], 5 * p2), (r["pdGap"], e["err_pdgap"])
    dense = e["wsum_is_dense"] if isinstance(e["wsum_is_dense"], list) else [e["wsum_is_dense"]]
    if all(x == 0 for x in dense) and e["reopt_rounds"] == 0:
        assert abs(r["pObj"] - e["pObj"]) <= 1e-9 * (1 + abs(e["pObj"]))
        assert abs(r["dObj"] - e["dObj"]) <= 1e-9 * (1 + abs(e["dObj"]))
        assert int(r["alm_inner"]) == int(e["alm_inner"]), (r["alm_inner"], e["alm_inner"])
        assert int(r["admm_iter"]) == int(e["admm_iter"]), (r["admm_iter"], e["admm_iter"])
        assert int(r["cg_iter"]) == int(e["admm_cg_iter"]), (r["cg_iter"], e["admm_cg_iter"])


def test_linearity_and_symmetry_of_operator_fullsize(built):
    """size-independent properties at a BASELINE-size block (n = 20000, r = 40): the CG operator is
    linear and self-adjoint, so <y, A x> == <x, A y>; checked through the public path by solving with
    tolerance 0 for one iteration is not possible, so we use the objective/constraint maps:
    A(sym(UV^T)) is bilinear and symmetric in (U,V)."""
    path = _gen("rand20000")
    s = common.hip_session(path, timesLogRank=4.0)
    try:
        info = s.block_info(0)
        assert info["rank"] == 40 and info["n"] == 20000
        rng = np.random.default_rng(5)
        U = rng.standard_normal((20000, 40))
        V = rng.standard_normal((20000, 40))
        be = s.be

        def A(X, Y):
            be.set_mat(host.MAT_U, 0, X)
            be.set_mat(host.MAT_V, 0, Y)
            be.init_constr(host.PAIR_UV)
            return be.get_vec(host.VEC_CONSTR_SUM)

        a_uv, a_vu = A(U, V), A(V, U)
        assert np.allclose(a_uv, a_vu, rtol=0, atol=1e-11 * np.abs(a_uv).max())   # symmetry
        a_2uv = A(2 * U, V)
        assert np.allclose(a_2uv, 2 * a_uv, rtol=0, atol=1e-11 * np.abs(a_uv).max())  # homogeneity
        a_sum = A(U + V, V)
        a_vv = A(V, V)
        assert np.allclose(a_sum, a_uv + a_vv, rtol=0, atol=1e-10 * np.abs(a_sum).max())  # additivity
        # objective of R = (U+V)/2 is a quadratic form: obj(2R) = 4 obj(R)
        be.set_mat(host.MAT_U, 0, U)
        be.set_mat(host.MAT_V, 0, V)
        o1 = be.cal_obj(host.PAIR_UV)
        be.set_mat(host.MAT_U, 0, 2 * U)
        be.set_mat(host.MAT_V, 0, 2 * V)
        o2 = be.cal_obj(host.PAIR_UV)
        assert np.isclose(o2, 4 * o1, rtol=1e-11)
    finally:
        s.close()


@pytest.mark.parametrize("name", ["maxcut100", "rand120", "blk4x60", "coupled3x70", "theta30"])
def test_fused_step_equals_separate_calls(built, name):
    """lorads_hip_admm_step (one sync, speculative enqueue with gates) must give exactly what the four
    separate entry points give, also when the speculation misses (first iterations, changing tol)."""
    g = common.golden_trace(name)
    sessions = [common.hip_session(common.instance_path(name)) for _ in range(2)]
    try:
        rank_warm = [int(x) for x in g["rank_warm"]]
        for s in sessions:
            if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                s.be.resize_rank(rank_warm)
            for k in range(s.nblk):
                n, r = s.block_shape(k)
                s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
            s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
            s.be.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
            s.be.cal_obj(host.PAIR_UV)
            s.be.update_dimacs(host.PAIR_UV)
        a, b = sessions
        rho = float(g["admm_rho"][0])
        for it, tol in enumerate([1e-8, 1e-8, 1e-12, 1e-6, 1e-14, 1e-9]):
            ca, pa, da, ea = a.be.admm_step(rho, tol, 800)
            cb = b.be.admm_update_var(rho, tol, 800)
            pb, db, eb = b.be.cal_obj(host.PAIR_UV), b.be.cal_dual_obj(), b.be.update_dimacs(host.PAIR_UV)
            assert ca == cb, (it, ca, cb)
            # b.lambda is summed over different workgroup partitions by the two paths (constraint-kernel partials vs
            # the flat dot of cal_dual_obj): equal to rounding; everything that feeds the iterates is identical
            assert pa == pb and ea == eb, (it, pa, pb, ea, eb)
            assert da == pytest.approx(db, rel=1e-14), (it, da, db)
            for k in range(a.nblk):
                assert np.array_equal(a.be.get_mat(host.MAT_U, k), b.be.get_mat(host.MAT_U, k))
                assert np.array_equal(a.be.get_mat(host.MAT_V, k), b.be.get_mat(host.MAT_V, k))
            a.be.update_dual_var(rho)
            b.be.update_dual_var(rho)
        # and against the reference's own numbers for the first iterations (tol as in the trace)
    finally:
        for s in sessions:
            s.close()


def test_cg_long_solve_with_restarts(built):
    """a solve that needs > 20 iterations exercises the k % 20 restart and the resume path; compared
    with the CPU oracle's CG (same quirks) on the same input"""
    path = common.instance_path("maxcut800")
    hs, os_ = _pair(path)
    try:
        rng = np.random.default_rng(3)
        n, r = hs.block_shape(0)
        U = rng.standard_normal((n, r))
        V = rng.standard_normal((n, r)) * 3
        for s in (hs, os_):
            s.be.set_mat(host.MAT_U, 0, U)
            s.be.set_mat(host.MAT_V, 0, V)
            s.be.set_vec(host.VEC_LAMBDA, np.linspace(-1, 1, s.m))
            s.be.init_constr(host.PAIR_UV)
        ia = hs.be.admm_update_var(0.3, 1e-9, 800)
        ib = os_.be.admm_update_var(0.3, 1e-9, 800)
        assert 42 < ib < 1500, ib
        assert abs(ia - ib) <= max(3, 0.05 * ib), (ia, ib)
        Ua, Ub = hs.be.get_mat(host.MAT_U, 0), os_.be.get_mat(host.MAT_U, 0)
        assert np.allclose(Ua, Ub, rtol=0, atol=1e-6 * np.abs(Ub).max())
    finally:
        hs.close()
        os_.close()


# ---------------------------------------------------------------- dual infeasibility
def _exact_min_eigs(prob, lam):
    return [float(np.linalg.eigvalsh(S.toarray())[0]) for S in common.slack_matrices(prob, lam)]


@pytest.mark.parametrize("name", TRACE_NAMES + ["densec300"])
def test_dual_infeasibility_vs_oracle_and_numpy(built, name):
    """lambda_min(C - A^*(lambda)) from the device Lanczos against numpy's dense eigensolver (exact) and the
    oracle slot, for arbitrary multipliers (indefinite slack).  Driven to 1e-10 it must agree to 1e-8; with
    the reference's ARPACK tolerance (1e-2) it must sit within that tolerance ABOVE the exact value (a Ritz
    value never undershoots)."""
    from lorads_amd import instances
    prob = instances.NAMED[name]()
    path = common.instance_path(name) if name != "densec300" else _gen(name)
    lam = np.random.default_rng(5).standard_normal(prob["m"])
    want, ex = common.exact_dual_infeasibility(prob, lam)
    hs, os_ = _pair(path)
    try:
        for s in (hs, os_):
            s.be.set_vec(host.VEC_LAMBDA, lam)
        tight, lam_min, _ = hs.hip_dual_infeasibility(tol=1e-10)
        assert np.allclose(lam_min, ex, rtol=1e-8, atol=1e-10)
        assert tight == pytest.approx(want, rel=1e-8)
        loose, lam_min2, nmv = hs.hip_dual_infeasibility()  # tol 1e-2, ncv 40, 600 restarts
        for a, e in zip(lam_min2, ex):
            assert e - 1e-9 * abs(e) <= a <= e + 1e-2 * abs(e)
        assert hs.be.dual_infeasibility() == pytest.approx(loose, rel=1e-12)
        assert os_.be.dual_infeasibility() == pytest.approx(tight, rel=1e-8)
        # through the host: the two divisions of data/lorads_solver.c:1034-1035
        assert hs.dual_infeasibility() == pytest.approx(loose / (1.0 + common.c_norm1(prob)), rel=1e-12)
    finally:
        hs.close(), os_.close()


@pytest.mark.parametrize("name,tlr", [("maxcut4000", 3.0), ("rand4000", 3.0), ("maxcut20000", 4.0), ("rand20000", 4.0)])
def test_dual_infeasibility_large_vs_arpack(built, name, tlr):
    """mid and BASELINE-size cones against ARPACK itself (scipy eigsh = dsaupd/dseupd) run to 1e-10 on the
    same slack matrix; the device result at the reference's tolerance must be within that tolerance, and the
    tight one equal to 1e-7."""
    import scipy.sparse.linalg as sla
    from lorads_amd import instances
    prob = instances.NAMED[name]()
    lam = np.random.default_rng(7).standard_normal(prob["m"])
    S = common.slack_matrices(prob, lam)[0]
    ref = float(sla.eigsh(S, k=1, which="SA", ncv=60, tol=1e-10, return_eigenvectors=False)[0])
    hs = common.hip_session(_gen(name), timesLogRank=tlr)
    try:
        hs.be.set_vec(host.VEC_LAMBDA, lam)
        _, lm_tight, nmv_t = hs.hip_dual_infeasibility(tol=1e-9)
        _, lm_loose, nmv_l = hs.hip_dual_infeasibility()
        assert lm_tight[0] == pytest.approx(ref, rel=1e-7)
        assert ref - 1e-9 * abs(ref) <= lm_loose[0] <= ref + 1e-2 * abs(ref)
        assert nmv_l <= nmv_t
    finally:
        hs.close()


def test_whole_solve_level2_hip_vs_oracle(built):
    """reoptLevel 2 (main.c:414-476): matcomp60 leaves phase 2 dual infeasible, so the extra round must run on
    both tables; both must end dual feasible with matching objectives."""
    res = []
    for mk in (common.hip_session, common.oracle_session):
        with mk(common.instance_path("matcomp60"), reoptLevel=2, phase1Tol=1e-2) as s:
            s.solve()
            res.append(s.results())
    h, o = res
    assert h["scale_obj_his"] == o["scale_obj_his"] == 5.0
    assert 0.0 <= h["dual_infeas_l1"] <= 5e-5 and 0.0 <= o["dual_infeas_l1"] <= 5e-5
    assert h["pObj"] == pytest.approx(o["pObj"], rel=1e-4)
    assert h["dObj"] == pytest.approx(o["dObj"], rel=1e-4)
    assert h["status"] == o["status"]


@pytest.mark.parametrize("name,params", [("maxcut100", dict(reoptLevel=0)), ("rand120", dict(reoptLevel=1, phase1Tol=1e-2)),
                                         ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2)),
                                         ("theta30", dict(reoptLevel=1, phase1Tol=1e-2))])
def test_fused_alm_step_equals_separate_calls(built, name, params):
    """alm_front/alm_step (one host round trip per inner iteration, next direction pre-computed) run the same
    kernels in the same order as the slot-by-slot path: the whole solve must be identical, iteration counts
    included."""
    res = []
    for fused in (1, 0):
        with common.hip_session(common.instance_path(name), **params) as s:
            assert s.be.has_alm_step
            s.use_fused_step(fused)
            s.solve()
            res.append(s.results())
    a, b = res
    for k in ("pObj", "constrVio1", "alm_outer", "alm_inner", "admm_iter", "cg_iter", "dual_infeas_l1"):
        assert a[k] == b[k], k
    # b.lambda is summed over different workgroup partitions by the fused evaluation and by cal_dual_obj
    for k in ("dObj", "pdGap"):
        assert a[k] == pytest.approx(b[k], rel=1e-12, abs=1e-14), k


def test_alm_front_and_step_function_level(built):
    """the two fused calls against the seven separate slots on the same state (rand120)"""
    path = common.instance_path("rand120")
    out = []
    for fused in (True, False):
        with common.hip_session(path) as s:
            be = s.be
            rho = 0.7
            be.init_constr(host.PAIR_RR)
            be.alm_cal_grad(rho)
            rec = []
            front = None
            for it in range(4):
                if fused:
                    p1, p2, coef = front if front is not None else be.alm_front(rho, it)
                else:
                    be.lbfgs_direction(it)
                    p1, p2 = be.alm_q12p12()
                    coef = be.alm_linesearch_coeffs(rho, p1, p2)
                tau, _ = common.linesearch_tau(coef)
                if fused:
                    lag, err1, np1, np2, ncoef = be.alm_step(rho, tau, it + 1)
                    front = (np1, np2, ncoef)
                else:
                    be.set_y_as_neg_grad()
                    be.alm_update_var(tau)
                    lag = be.alm_cal_grad(rho)
                    be.set_lbfgs_his_two(tau)
                    err1 = be.update_dimacs(host.PAIR_RR)
                rec.append([p1, p2, *coef, tau, lag, err1])
            rec.append(be.get_mat(host.MAT_R, 0).ravel().tolist())
            out.append(rec)
    for x, y in zip(*out):
        assert x == y


@pytest.mark.parametrize("name,grow", [("blk4x60", 0), ("mix4", 0), ("mix4", 13), ("blk4x60", 34)])
def test_lockstep_sweep_equals_cone_by_cone(built, name, grow):
    """Block-separable cones of equal rank are swept in lockstep on the merged cone (one launch chain for all cones,
    per-cone CG scalars).  Against the cone-by-cone sweep (LORADS_NO_BATCH) on the same state: same CG iteration
    counts per ADMM iteration, factors equal to rounding (the per-cone partial sums are taken in another order)."""
    g = common.golden_trace(name)
    path = common.instance_path(name)
    # (the lockstep sweep builds right-hand side and initial residual in two passes; the cone-by-cone sweep is asked to
    # do the same, so that what is compared is the sweep order and nothing else)
    os.environ["LORADS_SPLIT_FRONT"] = "1"
    try:
        sessions = [common.hip_session(path), common.hip_session(path)]
    finally:
        os.environ.pop("LORADS_SPLIT_FRONT", None)
    try:
        rank_warm = [int(x) for x in g["rank_warm"]]
        rng = np.random.default_rng(99)
        extra = {}
        for s in sessions:
            if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                s.be.resize_rank(rank_warm)
            if grow:  # AUG_RANK in mid-flight: the merged view, its chunk tables and the row kernels' shapes follow the rank
                s.be.resize_rank([grow] * s.nblk)
            for k in range(s.nblk):
                n, r = s.block_shape(k)
                R = g["R_warm_0_%d" % k].reshape(rank_warm[k], n).T
                if grow:
                    if k not in extra:
                        extra[k] = 0.05 * rng.standard_normal((n, grow - rank_warm[k]))
                    R = np.hstack([R, extra[k]])
                s.be.set_mat(host.MAT_R, k, R)
            s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
            s.be.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
            s.be.cal_obj(host.PAIR_UV)
            s.be.update_dimacs(host.PAIR_UV)
        a, b = sessions
        rho = float(g["admm_rho"][0])
        for it, tol in enumerate([1e-8, 1e-8, 1e-12, 1e-6, 1e-10, 1e-9]):
            os.environ.pop("LORADS_NO_BATCH", None)
            ca, pa, da, ea = a.be.admm_step(rho, tol, 800)
            os.environ["LORADS_NO_BATCH"] = "1"
            try:
                cb, pb, db, eb = b.be.admm_step(rho, tol, 800)
            finally:
                os.environ.pop("LORADS_NO_BATCH", None)
            assert ca == cb, (it, ca, cb)
            assert pa == pytest.approx(pb, rel=1e-11) and da == pytest.approx(db, rel=1e-11)
            assert ea == pytest.approx(eb, rel=1e-7, abs=1e-14)
            for k in range(a.nblk):
                for which in (host.MAT_U, host.MAT_V):
                    x, y = a.be.get_mat(which, k), b.be.get_mat(which, k)
                    assert np.allclose(x, y, rtol=0, atol=1e-10 * max(1.0, np.abs(y).max())), (it, k, which)
            a.be.update_dual_var(rho)
            b.be.update_dual_var(rho)
    finally:
        for s in sessions:
            s.close()


@pytest.mark.parametrize("name,ranks", [("rand120", [1, 2, 3, 7, 16, 33, 64, 66, 100, 129, 130, 200, 300]),
                                        ("maxcut100", [1, 5, 40, 65, 128, 131, 258]),
                                        ("matcomp60", [1, 4, 30, 67, 140]),
                                        ("blk4x60", [1, 6, 31, 70, 133]),
                                        ("densec40", [1, 2, 17, 48, 127, 128, 129, 144, 200, 258])])
def test_every_rank_shape_vs_oracle(built, name, ranks):
    """The row kernels are instantiated per (lanes per row, 16-byte loads, column steps): r even <= 128 -> 8 lanes x
    double2, odd or larger r -> 8 / 32 / 64 lanes x double.  Every shape, on the general (Gram), diagonal, single-entry,
    merged multi-cone and dense-C (MFMA; ranks above 128 take several launches of eight column tiles) operator paths, function
    by function against the oracle from the same random factors."""
    path = common.instance_path(name)
    for r in ranks:
        # the rank rule gives ceil(timesLogRank * ln n) (capped): aim at r from below, grow to r if the cap bites
        with common.oracle_session(path) as probe:
            n0 = probe.block_shape(0)[0]
        hs, os_ = _pair(path, timesLogRank=float((r - 0.5) / np.log(n0)))
        try:
            nb = hs.nblk
            cur = [hs.block_shape(k)[1] for k in range(nb)]
            assert max(cur) <= r, (cur, r)
            rng = np.random.default_rng(1000 + r)
            lam = 0.1 * rng.standard_normal(hs.m)
            if cur != [r] * nb:
                for s in (hs, os_):
                    s.be.resize_rank([r] * nb)
            rr = [hs.block_shape(k)[1] for k in range(nb)]
            Rs = [rng.standard_normal((hs.block_shape(k)[0], rr[k])) / np.sqrt(hs.block_shape(k)[0]) for k in range(nb)]
            for s in (hs, os_):
                for k in range(nb):
                    s.be.set_mat(host.MAT_R, k, Rs[k])
                s.be.set_vec(host.VEC_LAMBDA, lam)
            rho = 0.7
            vals = []
            for s in (hs, os_):
                be = s.be
                be.init_constr(host.PAIR_RR)
                lag = be.alm_cal_grad(rho)
                be.lbfgs_direction(0)
                p1, p2 = be.alm_q12p12()
                kk = be.alm_linesearch_coeffs(rho, p1, p2)
                tau, _ = common.linesearch_tau(kk)
                be.set_y_as_neg_grad()
                be.alm_update_var(min(tau, 0.5))
                lag2 = be.alm_cal_grad(rho)
                be.set_lbfgs_his_two(min(tau, 0.5))
                e1 = be.update_dimacs(host.PAIR_RR)
                be.lbfgs_direction(1)
                D = [be.get_mat(host.MAT_U, k) for k in range(nb)]
                po = be.cal_obj(host.PAIR_RR)
                be.alm_to_admm()
                be.init_constr(host.PAIR_UV)
                it = be.admm_update_var(1.5, 1e-9, 800)
                U = [be.get_mat(host.MAT_U, k) for k in range(nb)]
                V = [be.get_mat(host.MAT_V, k) for k in range(nb)]
                e2 = be.update_dimacs(host.PAIR_UV)
                vals.append((lag, p1, p2, kk, lag2, e1, D, po, it, U, V, e2))
            a, b = vals
            assert np.isclose(a[0], b[0], rtol=1e-10), (r, "lag")
            assert np.isclose(a[1], b[1], rtol=1e-9, atol=1e-9 * abs(b[2])) and np.isclose(a[2], b[2], rtol=1e-9), (r, "p12")
            assert np.allclose(a[3], b[3], rtol=1e-8, atol=1e-9 * max(abs(x) for x in b[3])), (r, "coef")
            assert np.isclose(a[4], b[4], rtol=1e-9) and np.isclose(a[5], b[5], rtol=1e-9), (r, "lag2/err")
            assert np.isclose(a[7], b[7], rtol=1e-10), (r, "obj")
            assert abs(a[8] - b[8]) <= max(2, 0.03 * b[8]), (r, "cg iterations", a[8], b[8])
            for k in range(nb):
                assert np.allclose(a[6][k], b[6][k], rtol=0, atol=1e-9 * np.abs(b[6][k]).max()), (r, k, "D")
                assert np.allclose(a[9][k], b[9][k], rtol=0, atol=2e-6 * np.abs(b[9][k]).max()), (r, k, "U")
                assert np.allclose(a[10][k], b[10][k], rtol=0, atol=2e-6 * np.abs(b[10][k]).max()), (r, k, "V")
            assert np.isclose(a[11], b[11], rtol=1e-4, atol=1e-9), (r, "err2")
        finally:
            hs.close()
            os_.close()


@pytest.mark.parametrize("name", ["rand120", "coupled3x70", "theta30", "mix4", "densec40", "sdplp40"])
def test_constraint_wise_operator_vs_reference_golden(built, name):
    """The constraint-wise operator (k_cw: one wavefront per constraint forms A_i(sym(x V^T)) straight from the factors;
    k_spmm<CW>: slot coefficient a * w_i) is chosen automatically for cones with >= 256 moderate-size constraints (the
    headline configuration); here it is forced on the small golden instances and replayed against the reference's
    vectors, function by function, like test_trace_vs_reference_golden."""
    os.environ["LORADS_OP_CW"] = "1"
    try:
        g = common.golden_trace(name)
        s = common.hip_session(common.instance_path(name))
        try:
            kinds = {s.hip_operator_kind(k) for k in range(s.nblk)}
            assert "k_cw+k_spmm_ell" in kinds, kinds
            log = common.replay_trace(s, g, rtol=1e-9, resync=True)
            assert len(log) > 50
        finally:
            s.close()
    finally:
        os.environ.pop("LORADS_OP_CW", None)


@pytest.mark.parametrize("name", ["mix4", "blk4x60", "sdplp40", "densec40"])
def test_objective_scaling_reaches_every_copy(built, name):
    """objScale_dualvar (data/lorads_solver.c:1040-1052) scales C and lambda in place; the device keeps C in several
    images (per-cone lists, union-pattern base, dense matrix, LP costs, the merged multi-cone view): after scale_obj
    every evaluation must agree with the oracle -- a stale copy showed up as a 5x (125x after three reopt rounds) too
    small objective on separable multi-cone problems."""
    hs, os_ = _pair(common.instance_path(name))
    try:
        rng = np.random.default_rng(11)
        lam = 0.3 * rng.standard_normal(hs.m)
        Rs = [rng.standard_normal(hs.block_shape(k)) / 3 for k in range(hs.nblk)]
        vals = []
        for s in (hs, os_):
            be = s.be
            for k in range(s.nblk):
                be.set_mat(host.MAT_R, k, Rs[k])
            be.set_vec(host.VEC_LAMBDA, lam)
            be.scale_obj(5.0)
            be.init_constr(host.PAIR_RR)
            lag = be.alm_cal_grad(0.9)
            po = be.cal_obj(host.PAIR_RR)
            be.lbfgs_direction(0)
            p1, p2 = be.alm_q12p12()
            be.alm_to_admm()
            be.init_constr(host.PAIR_UV)
            if be.has_admm_step:
                _, po2, do2, _ = be.admm_step(1.3, 1e-11, 800)
            else:
                be.admm_update_var(1.3, 1e-11, 800)
                po2, do2 = be.cal_obj(host.PAIR_UV), be.cal_dual_obj()
                be.update_dimacs(host.PAIR_UV)
            di = be.dual_infeasibility()
            vals.append((lag, po, p1, p2, po2, do2, di, be.get_vec(host.VEC_LAMBDA)))
        a, b = vals
        for i, key in enumerate(("lagNormSq", "pObj(RR)", "p1", "p2", "pObj(UV)", "dObj")):
            # (the last two follow two inexact CG sweeps: equal to the CG tolerance, not to rounding)
            assert a[i] == pytest.approx(b[i], rel=1e-7 if i < 4 else 2e-5, abs=1e-9), key
        assert a[6] == pytest.approx(b[6], rel=2e-2, abs=1e-9), "dual infeasibility"
        assert np.allclose(a[7], 5.0 * lam) and np.allclose(b[7], 5.0 * lam)
    finally:
        hs.close()
        os_.close()


_SWEEP = [dict(dyrankLevel=0), dict(dyrankLevel=3), dict(highAccMode=1), dict(lbfgsListLength=4), dict(reoptLevel=2, phase1Tol=1e-2),
          dict(initRho=0.5), dict(timesLogRank=0.5), dict(timesLogRank=0.0), dict(phase2Tol=1e-7)]


@pytest.mark.parametrize("name", ["mix4", "blk4x60", "sdplp40", "rand120", "theta50"])
def test_whole_solve_option_sweep_vs_oracle(built, name):
    """whole solves through the reference's command-line options that steer the outer loops (rank growth, reopt
    rounds, high-accuracy mode, L-BFGS memory, start penalty, rank rule): both tables must end feasible at the same
    objective.  (Iteration counts may differ: decisions sit on thresholds that rounding can cross.)"""
    for params in _SWEEP:
        res = []
        for mk in (common.hip_session, common.oracle_session):
            with mk(common.instance_path(name), **params) as s:
                s.solve()
                res.append(s.results())
        h, o = res
        assert h["constrVio1"] <= 2e-5 and o["constrVio1"] <= 2e-5, (params, h["constrVio1"], o["constrVio1"])
        tol = max(2e-4, 10 * o["pdGap"], 10 * h["pdGap"])
        assert abs(h["pObj"] - o["pObj"]) <= tol * (1 + abs(o["pObj"])), (params, h["pObj"], o["pObj"])
        assert abs(h["dObj"] - o["dObj"]) <= 5 * tol * (1 + abs(o["dObj"])), (params, h["dObj"], o["dObj"])


def test_operator_variants_agree_fullsize(built):
    """BASELINE-size cone (n = 20000, r = 40, 5000 constraints): the constraint-wise operator (k_cw + k_spmm<CW>, chosen
    automatically here) and the pair-dot/Gram operator (k_pairdots + k_sgram + k_spmm) are two implementations of
    x + A_V^*(A_V x); the same ADMM sweeps through either must give the same factors (CG driven to 1e-12)."""
    path = _gen("rand20000")
    rng = np.random.default_rng(21)
    R0 = rng.standard_normal((20000, 40)) / np.sqrt(20000)
    lam0 = 0.05 * rng.standard_normal(5000)
    res = []
    for cw in ("1", "0"):
        os.environ["LORADS_OP_CW"] = cw
        try:
            s = common.hip_session(path, timesLogRank=4.0)
        finally:
            os.environ.pop("LORADS_OP_CW", None)
        try:
            assert s.hip_operator_kind(0) == ("k_cw+k_spmm_ell" if cw == "1" else "k_pairdots+k_sgram+k_spmm2")
            be = s.be
            be.set_mat(host.MAT_R, 0, R0)
            be.set_vec(host.VEC_LAMBDA, lam0)
            be.alm_to_admm()
            be.init_constr(host.PAIR_UV)
            its = []
            for _ in range(3):
                c, pobj, dobj, err1 = be.admm_step(50.0, 1e-12, 800)
                be.update_dual_var(50.0)
                its.append(c)
            res.append((its, pobj, dobj, err1, be.get_mat(host.MAT_U, 0), be.get_mat(host.MAT_V, 0)))
        finally:
            s.close()
    (ia, pa, da, ea, Ua, Va), (ib, pb, db, eb, Ub, Vb) = res
    assert all(abs(x - y) <= 2 for x, y in zip(ia, ib)), (ia, ib)
    assert pa == pytest.approx(pb, rel=1e-8) and da == pytest.approx(db, rel=1e-8) and ea == pytest.approx(eb, rel=1e-6)
    assert np.allclose(Ua, Ub, rtol=0, atol=1e-8 * np.abs(Ub).max())
    assert np.allclose(Va, Vb, rtol=0, atol=1e-8 * np.abs(Vb).max())


@pytest.mark.parametrize("name,steps", [("rand120", 40), ("coupled3x70", 12), ("mix4", 12), ("sdplp40", 12)])
def test_recurrence_and_fused_front_equal_two_pass_form(built, name, steps):
    """k_cw cones keep A(sym(U V^T)) current through the CG updates (w += alpha A(sym(p V^T)), the operator's own
    output) instead of re-gathering the factors after every solve, and build right-hand side + initial residual of a
    solve in one pass (k_spmm2<FRONT>).  Both are re-arrangements of the same arithmetic: against the form that follows
    the reference step by step (LORADS_EXACT_REFRESH=1 LORADS_SPLIT_FRONT=1: full constraint refresh after every solve,
    separate passes) the sweeps must agree to rounding -- same CG iteration counts, factors to 1e-9 of their scale --
    over enough sweeps to cross the periodic exact refresh (every 32nd), with tolerances that give 0, a few and > 20 CG
    iterations per solve (restart path)."""
    g = common.golden_trace(name)
    os.environ["LORADS_OP_CW"] = "1"
    res = []
    try:
        for exact in (False, True):
            if exact:
                os.environ["LORADS_EXACT_REFRESH"] = "1"
                os.environ["LORADS_SPLIT_FRONT"] = "1"
            try:
                s = common.hip_session(common.instance_path(name))
            finally:
                os.environ.pop("LORADS_EXACT_REFRESH", None)
                os.environ.pop("LORADS_SPLIT_FRONT", None)
            try:
                assert "k_cw+k_spmm_ell" in {s.hip_operator_kind(k) for k in range(s.nblk)}
                rank_warm = [int(x) for x in g["rank_warm"]]
                if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                    s.be.resize_rank(rank_warm)
                for k in range(s.nblk):
                    n, r = s.block_shape(k)
                    s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
                s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
                s.be.alm_to_admm()
                s.be.init_constr(host.PAIR_UV)
                rho = float(g["admm_rho"][0])
                log = []
                tols = [1e-8, 1e-3, 1e-13, 1e-6, 1e-1, 1e-10]
                for it in range(steps):
                    if it % 3 == 2:   # the slot-by-slot entry points take the same route
                        c = s.be.admm_update_var(rho, tols[it % len(tols)], 800)
                        p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
                    else:
                        c, p, d, e = s.be.admm_step(rho, tols[it % len(tols)], 800)
                    s.be.update_dual_var(rho)
                    log.append((c, p, d, e))
                res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)],
                            [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]))
            finally:
                s.close()
    finally:
        os.environ.pop("LORADS_OP_CW", None)
    (la, Ua, Va), (lb, Ub, Vb) = res
    assert max(c for c, _, _, _ in lb) > 20, "no solve took the restart path"
    for it, (x, y) in enumerate(zip(la, lb)):
        assert abs(x[0] - y[0]) <= max(1, 0.02 * y[0]), (it, "cg iterations", x[0], y[0])
        assert x[1] == pytest.approx(y[1], rel=1e-8, abs=1e-10), (it, "pObj", x[1], y[1])
        assert x[2] == pytest.approx(y[2], rel=1e-8, abs=1e-10), (it, "dObj", x[2], y[2])
        assert x[3] == pytest.approx(y[3], rel=1e-6, abs=1e-12), (it, "err1", x[3], y[3])
    for A, B in ((Ua, Ub), (Va, Vb)):
        for x, y in zip(A, B):
            assert np.allclose(x, y, rtol=0, atol=1e-8 * max(np.abs(y).max(), 1e-300))


@pytest.mark.parametrize("name,steps,tlr", [("rand120", 40, None), ("coupled3x70", 12, None), ("sdplp40", 12, None), ("rand4000", 12, 3.0)])
def test_one_kernel_front_equals_coefficient_pass_plus_fused_front(built, name, steps, tlr):
    """k_front_cw (right-hand side, initial residual AND the per-slot contributions of iteration 0's constraint weights in
    one kernel, coefficients formed on the fly from the m-vectors; k_wsum instead of iteration 0's k_cw) against the form
    it replaces (LORADS_FRONT_CW=0: k_sval, k_spmm2<FRONT>, k_cw): the same sums in another order.  Same CG iteration
    counts, objectives to 1e-9, factors to 1e-9 of their scale, over sweeps that take 0, a few and > 20 CG iterations,
    through the fused step and through the slot-by-slot entry points, with the dual update riding on the front."""
    golden = tlr is None
    if golden:
        g = common.golden_trace(name)
        os.environ["LORADS_OP_CW"] = "1"
    res = []
    try:
        for knob in ("1", "0"):
            os.environ["LORADS_FRONT_CW"] = knob
            try:
                s = common.hip_session(common.instance_path(name)) if golden else common.hip_session(_gen(name), timesLogRank=tlr, phase1Tol=1e-2)
            finally:
                os.environ.pop("LORADS_FRONT_CW", None)
            try:
                assert "k_cw+k_spmm_ell" in {s.hip_operator_kind(k) for k in range(s.nblk)}
                if golden:
                    rank_warm = [int(x) for x in g["rank_warm"]]
                    if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                        s.be.resize_rank(rank_warm)
                    for k in range(s.nblk):
                        n, r = s.block_shape(k)
                        s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
                    s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
                    s.be.alm_to_admm()
                    rho = float(g["admm_rho"][0])
                else:
                    s.alm()
                    s.alm_to_admm()
                    rs = s.results()
                    rho = min(rs["admm_rho"] if rs["admm_rho"] > 0 else rs["alm_rho"], 5000.0)
                s.be.init_constr(host.PAIR_UV)
                log = []
                tols = [1e-8, 1e-3, 1e-13, 1e-6, 1e-1, 1e-10]
                for it in range(steps):
                    if it % 3 == 2:
                        c = s.be.admm_update_var(rho, tols[it % len(tols)], 800)
                        p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
                    else:
                        c, p, d, e = s.be.admm_step(rho, tols[it % len(tols)], 800)
                    s.be.update_dual_var(rho)
                    log.append((c, p, d, e))
                res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)],
                            s.be.get_vec(host.VEC_LAMBDA)))
            finally:
                s.close()
    finally:
        os.environ.pop("LORADS_OP_CW", None)
    (la, Ua, Va, lama), (lb, Ub, Vb, lamb) = res
    for it, (x, y) in enumerate(zip(la, lb)):
        assert abs(x[0] - y[0]) <= max(1, 0.02 * y[0]), (it, "cg iterations", x[0], y[0])
        assert x[1] == pytest.approx(y[1], rel=1e-9, abs=1e-11), (it, "pObj", x[1], y[1])
        assert x[2] == pytest.approx(y[2], rel=1e-9, abs=1e-11), (it, "dObj", x[2], y[2])
        assert x[3] == pytest.approx(y[3], rel=1e-6, abs=1e-12), (it, "err1", x[3], y[3])
    assert np.allclose(lama, lamb, rtol=0, atol=1e-9 * max(np.abs(lamb).max(), 1e-300))
    for A, B in ((Ua, Ub), (Va, Vb)):
        for x, y in zip(A, B):
            assert np.allclose(x, y, rtol=0, atol=1e-9 * max(np.abs(y).max(), 1e-300))


@pytest.mark.parametrize("name,cw,nobatch", [("rand120", "1", False), ("maxcut100", None, False), ("coupled3x70", "1", False),
                                             ("mix4", None, False), ("mix4", None, True), ("blk4x60", None, True),
                                             ("sdplp40", "1", False)])
def test_scalar_steps_on_carrier_kernels_are_bitwise_the_separate_launches(built, name, cw, nobatch):
    """The start of a CG solve (k_cg_init) and the convergence test after an update (k_cg_check) normally ride on a
    neighbouring kernel (k_cw; k_cg_dir / k_refresh_w / k_average): every workgroup re-derives the scalars from the same
    partial sums in the same order, workgroup 0 publishes them.  LORADS_LAZY_SCALARS=0 launches them as the
    one-workgroup kernels they replace.  Same arithmetic, so the sweeps must agree BIT FOR BIT -- iteration counts,
    objectives, factors -- including speculation misses and solves of 0, few and > 20 iterations."""
    g = common.golden_trace(name) if name != "maxcut100" else None
    if cw:
        os.environ["LORADS_OP_CW"] = cw
    if nobatch:  # cone-by-cone sweep: Max-Cut cones as stages > 0 (k_op_diag / k_cg_dir as carriers behind a gate)
        os.environ["LORADS_NO_BATCH"] = "1"
    res = []
    try:
        for lazy in ("1", "0"):
            os.environ["LORADS_LAZY_SCALARS"] = lazy
            try:
                s = common.hip_session(common.instance_path(name), phase1Tol=1e-2)
            finally:
                os.environ.pop("LORADS_LAZY_SCALARS", None)
            try:
                if g is not None:
                    rank_warm = [int(x) for x in g["rank_warm"]]
                    if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                        s.be.resize_rank(rank_warm)
                    for k in range(s.nblk):
                        n, r = s.block_shape(k)
                        s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
                    s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
                    rho = float(g["admm_rho"][0])
                else:
                    rng = np.random.default_rng(11)
                    n, r = s.block_shape(0)
                    s.be.set_mat(host.MAT_R, 0, rng.standard_normal((n, r)))
                    s.be.set_vec(host.VEC_LAMBDA, np.linspace(-1, 1, s.m))
                    rho = 0.3
                s.be.alm_to_admm()
                s.be.init_constr(host.PAIR_UV)
                log = []
                tols = [1e-8, 1e-3, 1e-13, 1e-6, 1e-1, 1e-10]
                for it in range(12):
                    if it % 3 == 2:
                        c = s.be.admm_update_var(rho, tols[it % len(tols)], 800)
                        p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
                    else:
                        c, p, d, e = s.be.admm_step(rho, tols[it % len(tols)], 800)
                    s.be.update_dual_var(rho)
                    log.append((c, p, d, e))
                res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)],
                            [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]))
            finally:
                s.close()
    finally:
        os.environ.pop("LORADS_OP_CW", None)
        os.environ.pop("LORADS_NO_BATCH", None)
    (la, Ua, Va), (lb, Ub, Vb) = res
    assert la == lb, (la, lb)
    for x, y in zip(Ua + Va, Ub + Vb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("name,tlr", [("blk4x60", None), ("mix4", None), ("blk16x4000", 2.0)])
def test_lockstep_convergence_test_on_the_last_workgroup_is_bitwise_its_own_launch(built, monkeypatch, name, tlr):
    """Lockstep sweep: the per-cone convergence test between two CG iterations as (a) every workgroup's own redo at the head of the
    next operator kernel, states written by each cone's first row tile, r.r in alternating slots (LORADS_SEG_CARRY, the default where
    the merged cone is of Max-Cut type); (c) the one-workgroup kernel k_cg_check_seg.  (Rounds 2-3 had a form (b), the test on the
    last workgroup of k_cg_update_seg to finish: bit-for-bit, slower twice, retired in round 4 with LORADS_SEG_LASTBLOCK -- the test
    keeps its name.)  Same sums in the same order: iterates, iteration counts and evaluations must be bit-for-bit equal -- also on
    cfg4 at full size (16 cones of n = 4000: 544 workgroups per update, every XCD involved)."""
    if name == "blk16x4000":
        path = os.path.join("/tmp", "lorads_test_blk16x4000.dat-s")
        if not os.path.exists(path):
            from lorads_amd import instances
            instances.write_sdpa(instances.NAMED["blk16x4000"](), path)
    else:
        path = common.instance_path(name)
    res = []
    # (carried by the next operator kernel -- the default; on the update's last workgroup; a launch of its own)
    # (... ; the same with the k % 20 == 0 restart's test and scalars as launches of their own -- round 2's form)
    # (... ; and with the start of the solves as k_cg_init_seg instead of on iteration 0's operator kernel)
    for carry, restart, init in (("1", "1", "1"), ("0", "1", "1"), ("1", "0", "1"), ("1", "1", "0"), ("1", "0", "0")):
        # (the last variant also stores the refresh after the U-solves -- k_pairdots + k_cv -- instead of letting the V front form its
        # weights from U_p . V_p itself)
        monkeypatch.setenv("LORADS_SEG_VIRT", "0" if (restart, init) == ("0", "0") else "1")
        monkeypatch.setenv("LORADS_SEG_CARRY_DUAL", "0" if (restart, init) == ("0", "0") else "1")      # (... and the dual update as k_dual_update)
        monkeypatch.setenv("LORADS_SEG_CARRY", carry)
        monkeypatch.setenv("LORADS_SEG_CARRY_RESTART", restart)
        monkeypatch.setenv("LORADS_SEG_CARRY_INIT", init)
        params = dict(phase1Tol=1e-1) if tlr is None else dict(phase1Tol=1e-1, timesLogRank=tlr)
        s = common.hip_session(path, **params)
        try:
            s.alm()
            s.alm_to_admm()
            r0 = s.results()
            rho = min(r0["admm_rho"] if r0["admm_rho"] > 0 else r0["alm_rho"], 5000.0)
            s.be.init_constr(host.PAIR_UV)
            log = []
            for tol in (1e-6, 1e-9, 1e-4, 1e-12, 1e-8, 1e-8):
                c, p_, d, e = s.be.admm_step(rho, tol, 300)
                s.be.update_dual_var(rho)
                log.append((c, p_, d, e))
            res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]))
        finally:
            s.close()
    (la, Ua, Va) = res[0]
    print(name, "CG iterations per step", [c for c, _, _, _ in la])
    assert max(c for c, _, _, _ in la) > 2 * len(Ua)
    for (lb, Ub, Vb) in res[1:]:
        assert la == lb, (la, lb)
        for x, y in zip(Ua + Va, Ub + Vb):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["blk4x60", "mix4"])
def test_lockstep_carriers_at_iteration_limits_and_later_restarts(built, monkeypatch, name):
    """The lockstep sweep's carried scalar steps where the plain runs of the test above do not go: iteration limits of 1, 2 and 3 (the
    solves' start rides on the only operator kernel there is; the k = 0 restart is the last thing a solve does), and solves that run
    to a limit of 45 at a tolerance nobody reaches -- the restarts at k = 20 and k = 40 then find a convergence test waiting and take
    it along on their residual pass.  Carried (default) against every scalar step as a launch of its own: bit for bit."""
    path = common.instance_path(name)
    res = []
    for carry in ("1", "0"):
        monkeypatch.setenv("LORADS_SEG_CARRY", carry)
        s = common.hip_session(path, phase1Tol=1e-1)
        try:
            s.alm()
            s.alm_to_admm()
            r0 = s.results()
            rho = min(r0["admm_rho"] if r0["admm_rho"] > 0 else r0["alm_rho"], 5000.0)
            s.be.init_constr(host.PAIR_UV)
            log = []
            for tol, maxit in ((1e-6, 1), (1e-6, 2), (1e-9, 3), (1e-30, 45), (1e-30, 45), (1e-6, 1), (1e-30, 21), (1e-8, 300)):
                c, p_, d, e = s.be.admm_step(rho, tol, maxit)
                s.be.update_dual_var(rho)
                log.append((c, p_, d, e))
            res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]))
        finally:
            s.close()
    (la, Ua, Va), (lb, Ub, Vb) = res
    print(name, "CG iterations per step", [c for c, _, _, _ in la])
    nb = len(Ua)
    assert la[0][0] == 2 * nb and la[3][0] == 2 * 45 * nb and la[6][0] == 2 * 21 * nb      # (every solve ran to its limit)
    assert la == lb, (la, lb)
    for x, y in zip(Ua + Va, Ub + Vb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("name,tlr", [("maxcut800", 1.9), ("rand4000", 2.9), ("blk4x60", None), ("mix4", None), ("densec40", None), ("sdplp40", None)])
def test_odd_ranks_run_padded_to_even_and_agree_with_the_unpadded_run(built, monkeypatch, name, tlr):
    """A cone of odd rank runs as the next even rank with a zero column in every factor (Block::rl / r: 16-byte row accesses and the
    kernels that need them for every rank) -- against the same work at the rank as it is (LORADS_PAD_ODD_RANK=0).  What the caller
    sees has the cone's own rank: set / get round trips, rank growth by an odd and by an even number of columns (old columns kept,
    new ones as AUG_RANK writes them), the first phase-1 gradient; then whole solves, which must end at the same objectives with the
    same amount of work up to the order of the row sums."""
    path = common.instance_path(name) if name not in ("rand4000",) else _gen(name)
    res = []
    for pad in ("1", "0"):
        monkeypatch.setenv("LORADS_PAD_ODD_RANK", pad)
        params = dict(phase1Tol=1e-3) if tlr is None else dict(phase1Tol=1e-3, timesLogRank=tlr)
        s = common.hip_session(path, **params)
        try:
            ranks0 = [s.block_shape(k)[1] for k in range(s.nblk)]
            assert any(r % 2 == 1 for r in ranks0), ranks0                   # (the case is about odd ranks)
            assert [s.hip_block_image(k)["rank"] for k in range(s.nblk)] == ranks0
            U0 = s.be.get_mat(host.MAT_U, 0)
            assert U0.shape[1] == ranks0[0]
            s.be.set_mat(host.MAT_U, 0, U0)
            assert np.array_equal(s.be.get_mat(host.MAT_U, 0), U0)
            grown = [min(r + (3 if k % 2 == 0 else 2), s.block_shape(k)[0]) for k, r in enumerate(ranks0)]
            Rpre = [s.be.get_mat(host.MAT_R, k) for k in range(s.nblk)]
            s.be.resize_rank(grown)
            assert [s.hip_block_image(k)["rank"] for k in range(s.nblk)] == grown
            for k in range(s.nblk):
                R = s.be.get_mat(host.MAT_R, k)
                n, r = Rpre[k].shape
                assert R.shape == (n, grown[k]) and np.array_equal(R[:, :r], Rpre[k])
                rr = min(n, grown[k] - r)
                want = np.zeros((n, grown[k] - r))
                want[np.arange(rr), np.arange(rr)] = 1 / np.sqrt(rr)
                assert np.array_equal(R[:, r:], want)
            s.be.init_constr(host.PAIR_RR)
            g0 = s.be.alm_cal_grad(1.0)
        finally:
            s.close()
        s = common.hip_session(path, **params)     # ... and a whole solve from the start point
        try:
            s.solve()
            res.append((g0, s.results()))
        finally:
            s.close()
    (ga, a), (gb, b) = res
    print(name, "first gradient", ga, gb, "alm inner", a["alm_inner"], b["alm_inner"], "admm", a["admm_iter"], b["admm_iter"], "pObj", a["pObj"], b["pObj"])
    assert ga == pytest.approx(gb, rel=1e-12)
    # (two solves that differ in rounding stop at the same tolerance, not at the same digits: phase2Tol = 1e-5)
    assert a["pObj"] == pytest.approx(b["pObj"], rel=1e-4) and a["dObj"] == pytest.approx(b["dObj"], rel=1e-4)
    assert a["alm_inner"] == pytest.approx(b["alm_inner"], rel=0.1, abs=3) and a["admm_iter"] == pytest.approx(b["admm_iter"], rel=0.1, abs=3)


@pytest.mark.parametrize("name,tlr", [("matcomp60", None), ("matcomp4000", 3.0)])
def test_bipartite_entry_operator_equals_the_one_kernel_form(built, monkeypatch, name, tlr):
    """Single-entry cones whose entry graph is bipartite (matrix completion): k_op_entry_bip forms every entry's pair dot once, on
    the rows of one colour, and hands it to the rows of the other colour (3 gathered rows per entry instead of 4).  Same sums up
    to the order of two additions: iterates to 1e-10 of scale, equal CG iteration counts, against k_op_entry (LORADS_ENTRY_BIP=0)."""
    path = common.instance_path(name) if name == "matcomp60" else _gen(name)
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("LORADS_ENTRY_BIP", on)
        params = dict(phase1Tol=1e-1) if tlr is None else dict(phase1Tol=1e-1, timesLogRank=tlr)
        s = common.hip_session(path, **params)
        try:
            assert s.hip_operator_kind(0) == ("k_op_entry_bip+k_op_entry_bip" if on == "1" else "k_op_entry")
            s.alm()
            s.alm_to_admm()
            r0 = s.results()
            rho = min(r0["admm_rho"] if r0["admm_rho"] > 0 else r0["alm_rho"], 5000.0)
            s.be.init_constr(host.PAIR_UV)
            log = []
            for tol in (1e-6, 1e-9, 1e-4, 1e-10):
                c, p_, d, e = s.be.admm_step(rho, tol, 300)
                s.be.update_dual_var(rho)
                log.append((c, p_, d, e))
            res.append((log, s.be.get_mat(host.MAT_U, 0), s.be.get_mat(host.MAT_V, 0)))
        finally:
            s.close()
    (la, Ua, Va), (lb, Ub, Vb) = res
    print(name, "CG iterations per step", [c for c, _, _, _ in la], [c for c, _, _, _ in lb])
    assert [c for c, _, _, _ in la] == [c for c, _, _, _ in lb]
    assert max(c for c, _, _, _ in la) > 4
    for (_, pa, da, ea), (_, pb, db, eb) in zip(la, lb):
        assert np.isclose(pa, pb, rtol=1e-10) and np.isclose(da, db, rtol=1e-10) and np.isclose(ea, eb, rtol=1e-7, atol=1e-14)
    assert np.allclose(Ua, Ub, rtol=0, atol=1e-10 * np.abs(Ub).max()) and np.allclose(Va, Vb, rtol=0, atol=1e-10 * np.abs(Vb).max())


def test_carried_scalar_steps_on_a_grid_larger_than_the_device(built):
    """A carried scalar step sums partials at the top of its carrier; the carrier writes its own partials at its end.
    With more workgroups than the device holds at once (Max-Cut n = 20000, r = 40: 625 resident of 625; n = 48000: 1500
    workgroups) late workgroups start after early ones have finished, so the two sets of partials must never share a
    slot -- they once did (start of a solve / restart residual vs the operator's p.Q partials), which showed up as a
    one-in-eight flake on a two-workgroup cone.  Bit-for-bit against the separate launches at n = 48000."""
    path = os.path.join("/tmp", "lorads_test_maxcut48000.dat-s")
    if not os.path.exists(path):
        from lorads_amd import instances
        instances.write_sdpa(instances.maxcut(48000, 200000, 48001), path)
    res = []
    for lazy in ("1", "0"):
        os.environ["LORADS_LAZY_SCALARS"] = lazy
        try:
            s = common.hip_session(path, timesLogRank=3.7, phase1Tol=1e-1)
        finally:
            os.environ.pop("LORADS_LAZY_SCALARS", None)
        try:
            s.alm()
            s.alm_to_admm()
            res0 = s.results()
            rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
            s.be.init_constr(host.PAIR_UV)
            log = []
            for tol in (1e-6, 1e-9, 1e-4, 1e-12):
                c, p, d, e = s.be.admm_step(rho, tol, 200)
                s.be.update_dual_var(rho)
                log.append((c, p, d, e))
            res.append((log, s.be.get_mat(host.MAT_U, 0)))
        finally:
            s.close()
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert max(c for c, _, _, _ in res[0][0]) > 2
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("name,nobatch,knob", [("maxcut100", False, "LORADS_SPLIT_FRONT"), ("mix4", True, "LORADS_SPLIT_FRONT"),
                                               ("blk4x60", True, "LORADS_SPLIT_FRONT"), ("maxcut100", False, "LORADS_EXACT_REFRESH"),
                                               ("blk4x60", True, "LORADS_EXACT_REFRESH"), ("mix4", True, "LORADS_EXACT_REFRESH")])
def test_fused_front_of_maxcut_cones_equals_the_separate_passes(built, name, nobatch, knob):
    """Max-Cut-type cones: right-hand side and initial residual of a solve come from one kernel (k_spmm2<FRONT> with the
    row-local operator g_p (x_p . V_p) V_p in its epilogue) instead of k_spmm2 + k_op_diag.  Same arithmetic up to how
    the compiler contracts the multiply-adds of the merged epilogue: against LORADS_SPLIT_FRONT=1 the sweeps must agree
    to rounding -- same CG iteration counts, objectives to 1e-10, factors to 1e-9 of their scale.
    knob = LORADS_EXACT_REFRESH: the same comparison for the constraint refresh of Max-Cut cones, which is made of row dots
    U_p . V_p that every front leaves exact and the CG updates keep current (k_op_diag's p_p . V_p), against the full
    refresh (k_pairdots + k_cv) after every solve."""
    g = common.golden_trace(name) if name != "maxcut100" else None
    if nobatch:
        os.environ["LORADS_NO_BATCH"] = "1"
    res = []
    try:
        for split in ("0", "1"):
            os.environ[knob] = split
            try:
                s = common.hip_session(common.instance_path(name), phase1Tol=1e-2)
            finally:
                os.environ.pop(knob, None)
            try:
                if g is not None:
                    rank_warm = [int(x) for x in g["rank_warm"]]
                    if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
                        s.be.resize_rank(rank_warm)
                    for k in range(s.nblk):
                        n, r = s.block_shape(k)
                        s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
                    s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
                    rho = float(g["admm_rho"][0])
                    s.be.alm_to_admm()
                else:  # the solver's own phase 1 and hand-over give the warm start and rho (as bench.py does)
                    s.alm()
                    s.alm_to_admm()
                    res0 = s.results()
                    rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
                s.be.init_constr(host.PAIR_UV)
                log = []
                for it, tol in enumerate([1e-8, 1e-6, 1e-13, 1e-7, 1e-5, 1e-10, 1e-9, 1e-12]):
                    c, p, d, e = s.be.admm_step(rho, tol, 800)
                    s.be.update_dual_var(rho)
                    log.append((c, p, d, e))
                res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)],
                            [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)]))
            finally:
                s.close()
    finally:
        os.environ.pop("LORADS_NO_BATCH", None)
    (la, Ua, Va), (lb, Ub, Vb) = res
    for it, (x, y) in enumerate(zip(la, lb)):
        assert abs(x[0] - y[0]) <= max(1, 0.01 * y[0]), (it, "cg iterations", x[0], y[0])
        assert x[1] == pytest.approx(y[1], rel=1e-10, abs=1e-12), (it, "pObj", x[1], y[1])
        assert x[2] == pytest.approx(y[2], rel=1e-10, abs=1e-12), (it, "dObj", x[2], y[2])
        assert x[3] == pytest.approx(y[3], rel=1e-8, abs=1e-12), (it, "err1", x[3], y[3])
    for x, y in zip(Ua + Va, Ub + Vb):
        assert np.allclose(x, y, rtol=0, atol=1e-9 * max(np.abs(y).max(), 1e-300))


def test_set_mat_invalidates_cached_pair_values(built):
    """The constraint values of (U, V) are cached for the next solve's initial residual (per cone and on the merged
    cone).  Overwriting a factor through the ABI must drop them: two sweeps, U replaced in between, against the
    oracle doing the same."""
    for name in ("mix4", "rand120"):
        g = common.golden_trace(name)
        hs, os_ = _pair(common.instance_path(name))
        try:
            rng = np.random.default_rng(5)
            newU = [0.3 * rng.standard_normal(hs.block_shape(k)) for k in range(hs.nblk)]
            out = []
            for s in (hs, os_):
                for k in range(s.nblk):
                    n, r = s.block_shape(k)
                    s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(int(g["rank_warm"][k]), n).T[:, :r])
                s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
                s.be.alm_to_admm()
                s.be.init_constr(host.PAIR_UV)
                rho = float(g["admm_rho"][0])
                s.be.admm_update_var(rho, 1e-11, 800)   # leaves cached pair values of the final (U, V)
                for k in range(s.nblk):
                    s.be.set_mat(host.MAT_U, k, newU[k])  # ... which are stale now
                s.be.init_constr(host.PAIR_UV)
                s.be.admm_update_var(rho, 1e-11, 800)
                out.append([s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)] + [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)])
            for x, y in zip(*out):
                assert np.allclose(x, y, rtol=0, atol=1e-7 * max(1.0, np.abs(y).max()))
        finally:
            hs.close()
            os_.close()


def test_fullsize_admm_vs_compiled_reference(built):
    """BASELINE size (n = 20000, r = 40, m = 5000): the compiled reference (oracle/_ref, travels with the snapshot) and
    the device path run the same 8 ADMM iterations from the same (U, V, lambda) -- the state after the device's own
    phase 1 -- with the same rho and tolerance rule.  Objectives to 1e-10 relative, identical CG iteration count."""
    drv = os.path.join(common.ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref not built (it is compiled from /root/reference in the build container)")
    path = _gen("rand20000")
    s = common.hip_session(path, timesLogRank=4.0, phase1Tol=1e-2, reoptLevel=0)
    try:
        s.alm()
        s.alm_to_admm()
        be = s.be
        res = s.results()
        rho = min(res["admm_rho"] if res["admm_rho"] > 0 else res["alm_rho"], 5000.0)
        U, V, lam = be.get_mat(host.MAT_U, 0), be.get_mat(host.MAT_V, 0), be.get_vec(host.VEC_LAMBDA)
        state = "/tmp/lorads_test_state_%d.bin" % os.getpid()
        with open(state, "wb") as f:
            f.write(np.asfortranarray(U).tobytes(order="F"))
            f.write(np.asfortranarray(V).tobytes(order="F"))
            f.write(lam.tobytes())
        its = 8
        env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
        r = subprocess.run([drv, path, "admmbench", "-", "--timesLogRank", "4.0", "--rho", repr(float(rho)), "--uv", state, "--nADMM",
                            str(its)], env=env, capture_output=True, text=True, timeout=600)
        os.remove(state)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("@@REF_ADMM_BENCH")]
        assert r.returncode == 0 and line, r.stderr[-400:]
        kv = dict(x.split("=") for x in line[0].split()[1:])
        be.init_constr(host.PAIR_UV)
        be.cal_obj(host.PAIR_UV)
        e0 = be.update_dimacs(host.PAIR_UV)
        e1, cg, pobj, dobj = s.admm_steps(its, rho, e0)
        assert int(cg) == int(kv["cg_iters"])
        assert pobj == pytest.approx(float(kv["pObj"]), rel=1e-10)
        assert dobj == pytest.approx(float(kv["dObj"]), rel=1e-10)
        assert e1 == pytest.approx(float(kv["err1"]), rel=1e-5)
    finally:
        s.close()


@pytest.mark.skipif(os.environ.get("LORADS_SKIP_LONG_TESTS") == "1", reason="~20 s of reference CPU time per instance on the GPU box's host")
@pytest.mark.parametrize("name,tlr,shape", [("maxcut20000", 4.0, (20000, 40)), ("rand20000", 4.0, (20000, 40)),
                                            ("blk16x4000", 2.0, (4000, 17)), ("matcomp4000", 2.0, (4000, 17)),
                                            ("sdplp2000", 2.0, (2000, 16)), ("matcomp50000", 5.5, (50000, 60))])
def test_fullsize_trace_vs_compiled_reference(built, name, tlr, shape):
    """Every lorads_func slot at BASELINE size (cfg3a, cfg3b: n = 20000, r = 40; cfg4: 16 cones n = 4000, the merged-cone /
    lockstep path; cfg5: n = 50000, r = 60, m = 2e5) against vectors the compiled reference produces on the spot:
    oracle/_ref/ref_driver in `trace` mode (3 ALM inner iterations, the reference's own phase 1 as warm start,
    2 ADMM iterations), replayed through the C ABI exactly like the small golden traces.
    cfg5 needs the reference's 64-bit index build (oracle/Makefile ref64: the 32-bit default cannot read n = 50000) and
    skips the reference's own phase 1 between the two parts (hours on one core): its ADMM part starts from the state
    the three traced ALM iterations left -- the same functions on the same inputs."""
    wide = shape[0] * shape[0] > 2**31 - 1
    drv = os.path.join(common.ROOT, "oracle", "_ref", "ref_driver64" if wide else "ref_driver")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref not built")
    path = _gen(name)
    dump = "/tmp/lorads_test_trace_%d.bin" % os.getpid()
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", LORADS_REF_ALLOW_LP="1")
    if wide:
        env.update(MKL_INTERFACE_LAYER="ILP64", LORADS_REF_NO_WARM="1")
    # (cfg5 without the reference's phase 1: ONE ADMM iteration -- from that cold state the second one runs into the CG
    # iteration limit on both sides and pins nothing; iteration 0 exercises both solves, refreshes, objective, dual update)
    n_admm = 1 if wide else 2
    r = subprocess.run([drv, path, "trace", dump, "--nALM", "3", "--nADMM", str(n_admm), "--timesLogRank", repr(tlr), "--phase1Tol", "1e-2"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-400:]
    g = common.read_dump(dump)
    os.remove(dump)
    g["_nALM"] = np.array([3.0])
    g["_nADMM"] = np.array([float(n_admm)])
    s = common.hip_session(path, timesLogRank=tlr, phase1Tol=1e-2)
    try:
        assert s.block_shape(0) == shape
        log = common.replay_trace(s, g, rtol=1e-9, resync=True)
        worst = max(e for _, e in log if not _[0:2] == "cg")
        print(name, "records", len(log), "worst rel-to-scale error", worst)
    finally:
        s.close()


def test_cfg5_slots_vs_oracle_fullsize(built):
    """BASELINE cfg5 (matrix completion n = 50000, m = 2e5 single-entry constraints, r = 60: k_op_entry and the phase-1
    kernels at the size the config names) slot by slot against the CPU oracle from the same state: 3 ALM inner
    iterations, then 2 ADMM iterations with their CG solves.  Independent of the reference build (the oracle needs none);
    the oracle's cost here is O(P r) per slot: about a minute in all."""
    path = _gen("matcomp50000")
    hs, os_ = _pair(path, timesLogRank=5.5)
    try:
        assert hs.block_shape(0) == (50000, 60) and hs.m == 200000
        assert hs.hip_operator_kind(0) == "k_op_entry_bip+k_op_entry_bip"   # (the entry graph of a matrix completion is bipartite)
        rho = 0.5
        for it in range(3):
            vals = []
            for s in (hs, os_):
                be = s.be
                if it == 0:
                    be.init_constr(host.PAIR_RR)
                lag = be.alm_cal_grad(rho)
                be.lbfgs_direction(it)
                p1, p2 = be.alm_q12p12()
                k = be.alm_linesearch_coeffs(rho, p1, p2)
                vals.append((lag, p1, p2, k, be.get_mat(host.MAT_U, 0), be.get_vec(host.VEC_Q1), be.get_vec(host.VEC_Q2),
                             be.get_mat(host.MAT_GRAD, 0)))
            (la, p1a, p2a, ka, Da, q1a, q2a, Ga), (lb, p1b, p2b, kb, Db, q1b, q2b, Gb) = vals
            assert np.isclose(la, lb, rtol=1e-10)
            assert np.isclose(p1a, p1b, rtol=1e-9, atol=1e-9 * abs(p2b))
            assert np.isclose(p2a, p2b, rtol=1e-9)
            assert np.allclose(ka, kb, rtol=1e-8, atol=1e-9 * max(abs(x) for x in kb))
            assert np.allclose(Ga, Gb, rtol=0, atol=1e-10 * np.abs(Gb).max())
            assert np.allclose(Da, Db, rtol=0, atol=1e-9 * np.abs(Db).max())
            assert np.allclose(q1a, q1b, rtol=0, atol=1e-10 * np.abs(q1b).max())
            assert np.allclose(q2a, q2b, rtol=0, atol=1e-10 * np.abs(q2b).max())
            tau, _ = common.linesearch_tau(kb)
            for s in (hs, os_):
                be = s.be
                be.set_y_as_neg_grad()
                be.alm_update_var(tau)
                be.alm_cal_grad(rho)
                be.set_lbfgs_his_two(tau)
            ea, eb = hs.be.update_dimacs(host.PAIR_RR), os_.be.update_dimacs(host.PAIR_RR)
            assert np.isclose(ea, eb, rtol=1e-9)
            Ra, Rb = hs.be.get_mat(host.MAT_R, 0), os_.be.get_mat(host.MAT_R, 0)
            assert np.allclose(Ra, Rb, rtol=0, atol=1e-11 * np.abs(Rb).max())
            hs.be.set_mat(host.MAT_R, 0, Rb)  # identical states: errors do not compound
        for s in (hs, os_):
            s.be.update_dual_var(rho)
            s.be.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
        la, lb = hs.be.get_vec(host.VEC_LAMBDA), os_.be.get_vec(host.VEC_LAMBDA)
        assert np.allclose(la, lb, rtol=0, atol=1e-11 * np.abs(lb).max())
        rho2 = 2.0
        for it in range(2):
            ia = hs.be.admm_update_var(rho2, 1e-7, 800)
            ib = os_.be.admm_update_var(rho2, 1e-7, 800)
            assert abs(ia - ib) <= max(2, 0.03 * ib), (ia, ib)
            Ua, Ub = hs.be.get_mat(host.MAT_U, 0), os_.be.get_mat(host.MAT_U, 0)
            Va, Vb = hs.be.get_mat(host.MAT_V, 0), os_.be.get_mat(host.MAT_V, 0)
            assert np.allclose(Ua, Ub, rtol=0, atol=2e-6 * np.abs(Ub).max())
            assert np.allclose(Va, Vb, rtol=0, atol=2e-6 * np.abs(Vb).max())
            ca, cb = hs.be.get_vec(host.VEC_CONSTR_SUM), os_.be.get_vec(host.VEC_CONSTR_SUM)
            assert np.allclose(ca, cb, rtol=0, atol=2e-6 * np.abs(cb).max())
            pa, pb = hs.be.cal_obj(host.PAIR_UV), os_.be.cal_obj(host.PAIR_UV)
            assert np.isclose(pa, pb, rtol=1e-6)
            da, db = hs.be.cal_dual_obj(), os_.be.cal_dual_obj()
            assert np.isclose(da, db, rtol=1e-9)
            ea, eb = hs.be.update_dimacs(host.PAIR_UV), os_.be.update_dimacs(host.PAIR_UV)
            assert np.isclose(ea, eb, rtol=1e-4, atol=1e-9)
            hs.be.set_mat(host.MAT_U, 0, Ub)
            hs.be.set_mat(host.MAT_V, 0, Vb)
            hs.be.update_dimacs(host.PAIR_UV)  # constrVal / constrValSum of the synchronised pair (quirk Q1: A(R R^T))
            for s in (hs, os_):
                s.be.update_dual_var(rho2)
            la, lb = hs.be.get_vec(host.VEC_LAMBDA), os_.be.get_vec(host.VEC_LAMBDA)
            assert np.allclose(la, lb, rtol=0, atol=1e-9 * np.abs(lb).max())
            print("cfg5 ADMM iteration", it, "cg", ia, ib, "pObj", pa, pb, "err1", ea, eb)
    finally:
        hs.close()
        os_.close()


@pytest.mark.parametrize("name,tlr,nobatch", [("maxcut100", None, False), ("blk4x60", None, False), ("blk4x60", None, True),
                                              ("mix4", None, False), ("rand120", None, False), ("rand4000", 4.0, False),
                                              ("maxcut800", None, False), ("sdplp40", None, False)])
def test_graph_replay_is_bitwise_the_launch_by_launch_iteration(built, name, tlr, nobatch):
    """An ADMM iteration whose launch chain has been seen before is replayed as a captured hipGraph (graph.inc): rho, the CG
    tolerance, the iteration limit and the hand-over's sequence number live in device memory, so a replay runs exactly the
    kernels the enqueue would have launched, with the same arguments.  LORADS_GRAPH=0 enqueues every iteration launch by
    launch.  The two must agree BIT FOR BIT -- CG iteration counts, objectives, factors, multipliers -- over iterations that
    change rho and the tolerance, miss their speculation (resumed launch by launch), mix the fused step with the
    slot-by-slot calls, and cross the periodic exact constraint refresh (every 32nd sweep: a chain of its own)."""
    path = _gen(name) if name in ("rand4000",) else common.instance_path(name)
    if nobatch:
        os.environ["LORADS_NO_BATCH"] = "1"
    res = []
    try:
        for graph in ("2", "0"):   # (2: the lockstep sweep of merged cones is replayed too)
            os.environ["LORADS_GRAPH"] = graph
            try:
                kw = dict(phase1Tol=1e-2)
                if tlr:
                    kw["timesLogRank"] = tlr
                s = common.hip_session(path, **kw)
            finally:
                os.environ.pop("LORADS_GRAPH", None)
            try:
                s.alm()           # the solver's own phase 1 (the same launches either way: phase 1 is not replayed)
                s.alm_to_admm()
                s.be.init_constr(host.PAIR_UV)
                res0 = s.results()
                rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)   # (the hand-off's penalty, as bench.py)
                log = []
                tols = [1e-8, 1e-8, 1e-8, 1e-8, 1e-3, 1e-8, 1e-8, 1e-12, 1e-8, 1e-8]
                for it in range(70):
                    if it % 7 == 6:
                        c = s.be.admm_update_var(rho, tols[it % len(tols)], 800)
                        p, d, e = s.be.cal_obj(host.PAIR_UV), s.be.cal_dual_obj(), s.be.update_dimacs(host.PAIR_UV)
                    else:
                        c, p, d, e = s.be.admm_step(rho, tols[it % len(tols)], 800 if it != 33 else 3)
                    s.be.update_dual_var(rho)
                    if it % 10 == 9:
                        rho *= 1.2
                    log.append((c, p, d, e))
                assert all(np.isfinite(x[1]) and np.isfinite(x[3]) for x in log), log[-3:]
                st = s.hip_graph_stats()
                res.append((log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)],
                            [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)], s.be.get_vec(host.VEC_LAMBDA), st))
            finally:
                s.close()
    finally:
        os.environ.pop("LORADS_NO_BATCH", None)
    (la, Ua, Va, lama, sta), (lb, Ub, Vb, lamb, stb) = res
    # (most iterations of a single cone are replays; several cones whose CG counts wander independently return to a state less often)
    assert sta["enabled"] == 1 and sta["replayed"] >= (10 if len(Ua) == 1 else 3), sta
    assert stb["enabled"] == 0 and stb["captured"] == 0 and stb["replayed"] == 0, stb   # (... and none with the switch off)
    assert la == lb, [(i, x, y) for i, (x, y) in enumerate(zip(la, lb)) if x != y][:3]
    assert np.array_equal(lama, lamb)
    for x, y in zip(Ua + Va, Ub + Vb):
        assert np.array_equal(x, y)


def test_rank_growth_on_the_device_fullsize(built):
    """AUG_RANK (data/lorads_solver.c:806-906) at BASELINE cfg5's size, n = 50000, r = 60 -> 90, and on a multi-cone problem with
    padded rows: the old columns are kept, the new ones hold 1/sqrt(k) on their leading diagonal, in all four factors -- done by
    one kernel per factor and cone on the device (no copy through the host), so the call itself must be fast."""
    import time
    for name, tlr, grow in (("matcomp50000", 5.5, 1.5), ("mix4", None, 1.5)):
        path = _gen(name) if name == "matcomp50000" else common.instance_path(name)
        kw = dict(timesLogRank=tlr) if tlr else {}
        s = common.hip_session(path, **kw)
        try:
            rng = np.random.default_rng(3)
            shapes = [s.block_shape(k) for k in range(s.nblk)]
            mats = {}
            for which in (host.MAT_R, host.MAT_U, host.MAT_V, host.MAT_GRAD):
                for k, (n, r) in enumerate(shapes):
                    mats[which, k] = rng.standard_normal((n, r))
                    s.be.set_mat(which, k, mats[which, k])
            newr = [min(n, int(np.ceil(r * grow))) for n, r in shapes]
            s.hip_sync()
            t0 = time.perf_counter()
            s.be.resize_rank(newr)
            s.hip_sync()
            dt = time.perf_counter() - t0
            print(name, "resize_rank", [r for _, r in shapes], "->", newr, "%.2f ms" % (1e3 * dt))
            for which in (host.MAT_R, host.MAT_U, host.MAT_V, host.MAT_GRAD):
                for k, (n, r) in enumerate(shapes):
                    got = s.be.get_mat(which, k)
                    assert got.shape == (n, newr[k])
                    assert np.array_equal(got[:, :r], mats[which, k])
                    aug = newr[k] - r
                    want = np.zeros((n, aug))
                    rr = min(n, aug)
                    want[np.arange(rr), np.arange(rr)] = 1 / np.sqrt(rr)
                    assert np.array_equal(got[:, r:], want)
            if name == "matcomp50000":
                # 4 x 50000 x 90 doubles = 144 MB written + 96 MB read on the device: a few hundred microseconds of kernels; the
                # rest is allocation.  The host round trip this replaces moved 4 x 2 x 24 MB over PCIe and transposed them twice
                # on one host thread (~0.5 s).
                assert dt < 0.25, dt
            # the grown factors work: one evaluation and one sweep run through
            s.be.alm_to_admm()
            s.be.init_constr(host.PAIR_UV)
            c_, p_, d_, e_ = s.be.admm_step(1.0, 1e-6, 5)
            assert np.isfinite(p_) and np.isfinite(e_)
        finally:
            s.close()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,tlr,its", [("rand20000", 4.0, 6), ("maxcut20000", 4.0, 6), ("blk16x4000", 2.0, 5), ("matcomp50000", 5.5, 4),
                                          # the reference's DENSE branch (dsyr2k / dsymm round differently from any sparse order): cfg1's
                                          # look-alike and matcomp60 over 25 iterations from the same start -- north_star's 1e-6 pinned
                                          # on iterates that cannot wander to another stationary point, as whole solves of these two can
                                          ("theta50", None, 25), ("matcomp60", None, 25), ("densec40", None, 25)])
def test_fullsize_admm_iterations_from_the_devices_own_state_vs_compiled_reference(built, name, tlr, its):
    """BASELINE cfg3b / cfg3a / cfg4 / cfg5 at full size: the device runs its own phase 1, its (U, V, lambda) go to the COMPILED
    REFERENCE (oracle/_ref/ref_driver admmbench -- ref_driver64, the 64-bit index build, where n^2 > 2^31), which runs `its` ADMM
    iterations of LORADSUpdateSDPVar + objective + DIMACS + dual update from that state on the host; the device then replays the same
    iterations from the same state.  Same penalty, same tolerance rule.  Demanded: the same number of CG iterations, objectives equal
    to 1e-10 relative, err1 to the seven digits the driver prints (what bench.py reports as parity_full_size in its line, here as an
    assertion)."""
    import subprocess
    import bench
    n_by = {"rand20000": 20000, "maxcut20000": 20000, "blk16x4000": 4000, "matcomp50000": 50000}.get(name, 100)
    dense_branch = name in ("theta50", "matcomp60", "densec40")
    wide = n_by * n_by > 2**31 - 1
    drv = os.path.join(common.ROOT, "oracle", "_ref", "ref_driver64" if wide else "ref_driver")
    if not os.path.exists(drv):
        pytest.skip("compiled reference not present (oracle/_ref travels with the snapshot from the build container)")
    path = common.instance_path(name) if dense_branch else _gen(name)
    s = host.Session.open(path)
    s.set_params(verbose=0, phase1Tol=1e-2, reoptLevel=0)
    if tlr:
        s.set_params(timesLogRank=tlr)
    tlr = tlr or 2.0    # (the reference's default, main.c:29)
    if name == "matcomp50000":
        s.set_params(dyrankLevel=0)      # (r = 60 as BASELINE names it)
    s.prepare(1, 0)
    s.attach_hip()
    state = "/tmp/lorads_test_state_%d_%s.bin" % (os.getpid(), name)
    try:
        s.alm()
        s.alm_to_admm()
        res = s.results()
        rho = min(res["admm_rho"] if res["admm_rho"] > 0 else res["alm_rho"], 5000.0)
        be = s.be
        UV0 = [(be.get_mat(host.MAT_U, k), be.get_mat(host.MAT_V, k)) for k in range(s.nblk)]
        lam0 = be.get_vec(host.VEC_LAMBDA)
        with open(state, "wb") as f:
            for U, V in UV0:
                f.write(np.asfortranarray(U).tobytes(order="F"))
                f.write(np.asfortranarray(V).tobytes(order="F"))
            f.write(lam0.tobytes())
        env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1",
                   LORADS_REF_UV_RANKS=",".join(str(s.block_shape(k)[1]) for k in range(s.nblk)))
        if wide:
            env["MKL_INTERFACE_LAYER"] = "ILP64"
        r = subprocess.run([drv, path, "admmbench", "-", "--timesLogRank", repr(tlr), "--rho", repr(rho), "--uv", state, "--nADMM", str(its)],
                           env=env, capture_output=True, text=True, timeout=1400)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("@@REF_ADMM_BENCH")]
        assert r.returncode == 0 and line, r.stderr[-800:]
        ref = dict(x.split("=") for x in line[0].split()[1:])
        be.init_constr(host.PAIR_UV)
        be.cal_obj(host.PAIR_UV)
        e0 = be.update_dimacs(host.PAIR_UV)
        e1, cg1, p1, d1 = bench.admm_steps(be, host, rho, e0, its, s)
        print(name, "reference", ref, "device", dict(pObj=p1, dObj=d1, err1=e1, cg_iters=cg1))
        if dense_branch:   # (a CG that stops on the threshold may fall the other way under the dense branch's rounding)
            assert abs(int(cg1) - int(ref["cg_iters"])) <= max(3, 0.02 * int(ref["cg_iters"])), (cg1, ref["cg_iters"])
            assert p1 == pytest.approx(float(ref["pObj"]), rel=1e-8, abs=1e-9)      # (north_star asks 1e-6)
            assert d1 == pytest.approx(float(ref["dObj"]), rel=1e-8, abs=1e-9)
            assert e1 == pytest.approx(float(ref["err1"]), rel=1e-5, abs=1e-12)
            return
        assert int(cg1) == int(ref["cg_iters"]), (cg1, ref["cg_iters"])
        assert p1 == pytest.approx(float(ref["pObj"]), rel=1e-10, abs=1e-10)
        assert d1 == pytest.approx(float(ref["dObj"]), rel=1e-10, abs=1e-10)
        assert e1 == pytest.approx(float(ref["err1"]), rel=5e-7, abs=1e-14)   # (the driver prints seven digits of it)
    finally:
        s.close()
        if os.path.exists(state):
            os.remove(state)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,r", [("matcomp50000", 91), ("matcomp50000", 135), ("maxcut140000", 40), ("rand140000", 24)])
def test_large_cones_and_odd_ranks_are_not_refused(built, name, r):
    """The reference has no limit on a cone's dimension or rank (data/lorads_solver.c:290-319).  The row kernels leave one partial
    sum per workgroup, and their slots used to hold 4096: n = 50000 was refused at any odd rank above 64 (32 lanes per row: 8 rows per
    workgroup) and n = 140000 at every rank.  The slots are now sized to the context's largest cone.  Function by function against
    the oracle from the same factors (phase-1 gradient, direction, q1/q2, objective, DIMACS residual; three CG iterations of one
    ADMM sweep), at n = 50000 with r = 91 and r = 135 (rank growth from 60) and at n = 140000."""
    if name == "maxcut140000":
        instances.NAMED.setdefault(name, lambda: instances.maxcut(140000, 6 * 140000, 140000))
    if name == "rand140000":
        instances.NAMED.setdefault(name, lambda: instances.randsparse(140000, 35000, 140001, c_edges=6 * 140000))
    path = _gen(name)
    tlr = 5.5 if name == "matcomp50000" else (r - 0.5) / np.log(140000)
    hs, os_ = _pair(path, timesLogRank=float(tlr))
    try:
        n, r0 = hs.block_shape(0)
        if r0 != r:
            for s in (hs, os_):
                s.be.resize_rank([r])
        assert hs.block_shape(0) == (n, r) and os_.block_shape(0) == (n, r)
        rng = np.random.default_rng(r)
        R = rng.standard_normal((n, r)) / np.sqrt(n)
        lam = 0.1 * rng.standard_normal(hs.m)
        rho = 0.7
        vals = []
        for s in (hs, os_):
            be = s.be
            be.set_mat(host.MAT_R, 0, R)
            be.set_vec(host.VEC_LAMBDA, lam)
            be.init_constr(host.PAIR_RR)
            lag = be.alm_cal_grad(rho)
            G = be.get_mat(host.MAT_GRAD, 0)
            be.lbfgs_direction(0)
            p1, p2 = be.alm_q12p12()
            q1, q2 = be.get_vec(host.VEC_Q1), be.get_vec(host.VEC_Q2)
            po = be.cal_obj(host.PAIR_RR)
            e1 = be.update_dimacs(host.PAIR_RR)
            be.alm_to_admm()
            be.init_constr(host.PAIR_UV)
            it = be.admm_update_var(1.5, 1e-14, 3)       # (three CG iterations per solve: the limit, not the tolerance, stops them)
            vals.append((np.array([lag, p1, p2, po, e1]), G, q1, q2, be.get_mat(host.MAT_U, 0), be.get_mat(host.MAT_V, 0), it))
        a, b = vals
        assert np.allclose(a[0], b[0], rtol=1e-9), (a[0], b[0])
        assert np.allclose(a[1], b[1], rtol=0, atol=1e-10 * np.abs(b[1]).max())
        assert np.allclose(a[2], b[2], rtol=0, atol=1e-10 * np.abs(b[2]).max()) and np.allclose(a[3], b[3], rtol=0, atol=1e-10 * np.abs(b[3]).max())
        assert a[6] == b[6] == 6, (a[6], b[6])
        assert np.allclose(a[4], b[4], rtol=0, atol=1e-8 * np.abs(b[4]).max()) and np.allclose(a[5], b[5], rtol=0, atol=1e-8 * np.abs(b[5]).max())
    finally:
        hs.close()
        os_.close()


def _run_admm_steps(path, env, steps, **kw):
    """phase 1 (the solver's own) + `steps` ADMM iterations under the environment `env`; returns (log, U, V, lambda)"""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = common.hip_session(path, phase1Tol=1e-2, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        s.alm()
        s.alm_to_admm()
        s.be.init_constr(host.PAIR_UV)
        res0 = s.results()
        rho = min(res0["admm_rho"] if res0["admm_rho"] > 0 else res0["alm_rho"], 5000.0)
        log = []
        e = s.be.update_dimacs(host.PAIR_UV)
        for it in range(steps):
            c, p, d, e = s.be.admm_step(rho, min(1e-2 * e, 1e-8), 800)
            s.be.update_dual_var(rho)
            log.append((c, p, d, e))
        return (log, [s.be.get_mat(host.MAT_U, k) for k in range(s.nblk)], [s.be.get_mat(host.MAT_V, k) for k in range(s.nblk)],
                s.be.get_vec(host.VEC_LAMBDA), [s.hip_operator_kind(k) for k in range(s.nblk)])
    finally:
        s.close()


@pytest.mark.parametrize("name", ["rand120", "maxcut100", "blk4x60", "theta30"])
def test_gram_form_direction_on_one_gpu_replays_the_reference_trace(built, monkeypatch, name):
    """LORADS_LBFGS_GRAM=2: the L-BFGS direction of a single GPU in Gram form (one pass forms the 15 products of {Grad, y_t, s_t}, one
    launch sums them and runs the two-loop recursion on coefficients, one pass forms D: two passes over the vectors instead of five) --
    the sharded path's form, on one rank.  Against the reference's golden trace, and through a whole solve against the default form
    (the recursion): same objectives; the inner iteration counts agree where the instance is easy and drift where it is not -- which
    is why the recursion, the reference's own order of operations, stays the default on one GPU."""
    monkeypatch.setenv("LORADS_LBFGS_GRAM", "2")
    g = common.golden_trace(name)
    s = common.hip_session(common.instance_path(name))
    try:
        log = common.replay_trace(s, g, rtol=1e-9, resync=True)
        w = common.trace_worst(log)
        print(name, "Gram-form direction, worst rel-to-scale errors", w)
        assert w["phase1"] <= 1e-9, w
    finally:
        s.close()
    res = []
    for gram in ("2", "1"):
        monkeypatch.setenv("LORADS_LBFGS_GRAM", gram)
        with common.hip_session(common.instance_path(name), phase1Tol=1e-3) as s2:
            s2.solve()
            res.append(s2.results())
    a, b = res
    assert a["pObj"] == pytest.approx(b["pObj"], rel=1e-5) and a["dObj"] == pytest.approx(b["dObj"], rel=1e-5)
    print(name, "inner iterations: Gram form", a["alm_inner"], "recursion", b["alm_inner"])
    assert abs(a["alm_inner"] - b["alm_inner"]) <= max(5, (0.3 if name == "theta30" else 0.1) * b["alm_inner"]), (a["alm_inner"], b["alm_inner"])


@pytest.mark.parametrize("name,tlr", [("densea40", None), ("densea300", 2.0), ("denseac200", 3.0)])
def test_dense_constraint_products_kept_for_a_solve_equal_the_per_application_gemms(built, name, tlr):
    """Cones with dense constraint matrices: W_j = A_j V is formed once per CG solve (V is fixed for its length) and every operator
    application takes <A_j, sym(x V^T)> = <x, W_j> and (sum_j mu_j A_j) V = sum_j mu_j W_j from it -- nd GEMMs per solve instead of
    nd + 1 per application.  LORADS_DENSE_CACHE=0: the per-application form.  Same CG iteration counts, iterates equal to rounding;
    and the kept form is the faster one."""
    import time
    path = common.instance_path(name) if name == "densea40" else _gen(name)
    kw = dict(timesLogRank=tlr) if tlr else {}
    out = []
    for cache in ("1", "0"):
        t0 = time.perf_counter()
        r = _run_admm_steps(path, {"LORADS_DENSE_CACHE": cache}, 12, **kw)
        out.append((r, time.perf_counter() - t0))
    (a, ta), (b, tb) = out
    assert all("dense A_i" in k for k in a[4]), a[4]
    print(name, "12 ADMM iterations + phase 1: kept products %.3f s, per-application GEMMs %.3f s; CG iterations" % (ta, tb), [x[0] for x in a[0]][:6])
    for x, y in zip(a[0], b[0]):
        assert abs(x[0] - y[0]) <= max(2, 0.02 * y[0]), (x, y)
        assert x[1] == pytest.approx(y[1], rel=1e-8, abs=1e-10) and x[2] == pytest.approx(y[2], rel=1e-8, abs=1e-10)
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert np.allclose(x, y, rtol=0, atol=1e-7 * np.abs(y).max())


@pytest.mark.parametrize("name,tlr,knob", [("maxcut800", None, "LORADS_FRONT_DIAG"), ("maxcut800", None, "LORADS_EVAL_DIAG"),
                                           ("maxcut800", None, "LORADS_FUSE_DIR"), ("maxcut800", None, "LORADS_TILE_UPDATE"),
                                           ("rand4000", 4.0, "LORADS_FOLD_AVG"), ("rand4000", 4.0, "LORADS_FUSE_EVAL"),
                                           ("rand4000", 4.0, "LORADS_CW_QUAD")])
def test_fused_paths_equal_the_step_by_step_forms(built, name, tlr, knob):
    """Every re-arrangement of the launch chain keeps a switch that restores the form it replaces.  Those that no
    other test sets side by side: the Max-Cut front's on-the-fly diagonal coefficients, the one-kernel evaluation of Max-Cut cones,
    the direction update inside k_op_diag, k_cg_update on the operator's row tiles; the average folded into the last update, the
    evaluation in one launch and the four-lane k_cw of cones on the k_cw path.  Each switched off against the default over 25 ADMM
    iterations from the solver's own phase 1: same CG iteration counts (+-1 where a test sits on its threshold), objectives to
    1e-9, factors to 1e-8 of scale (same sums, other groupings)."""
    path = _gen(name) if name == "rand4000" else common.instance_path(name)
    kw = dict(timesLogRank=tlr) if tlr else {}
    a = _run_admm_steps(path, {}, 25, **kw)
    b = _run_admm_steps(path, {knob: "0"}, 25, **kw)
    for it, (x, y) in enumerate(zip(a[0], b[0])):
        assert abs(x[0] - y[0]) <= 1, (it, x, y)
        assert x[1] == pytest.approx(y[1], rel=1e-9, abs=1e-11) and x[2] == pytest.approx(y[2], rel=1e-9, abs=1e-11), (it, x, y)
    assert np.allclose(a[3], b[3], rtol=0, atol=1e-8 * max(np.abs(b[3]).max(), 1e-300))
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert np.allclose(x, y, rtol=0, atol=1e-8 * np.abs(y).max())
