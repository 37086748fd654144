"""Shared helpers of the parity tests: oracle loading, golden loading and the scripted call
sequence that oracle/ref_driver.c (mode `trace`) ran against the REFERENCE, replayed against any
operator table (CPU oracle or HIP)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

from lorads_amd import host  # noqa: E402

_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liblorads_oracle.so"))
        host._bind(lib)
        lib.lorads_oracle_backend_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(host.BackendStruct)]
        _oracle = lib
    return _oracle


def oracle_session(path, world=1, rank=0, separable=False, **params):
    """Session driven by the CPU oracle table (tests / cpu_baseline only)."""
    lib = load_oracle()
    s = host.Session.open(path, lib=lib)
    s.set_params(verbose=0, **params)
    s.prepare(world, rank, separable=separable)
    st = host.BackendStruct()
    assert lib.lorads_oracle_backend_create(s.problem_ptr(), 2, C.byref(st)) == 0
    s.attach(st)
    return s


def hip_session(path, world=1, rank=0, separable=None, **params):
    s = host.Session.open(path)
    s.set_params(verbose=0, **params)
    # (sharded: a block-separable deal runs on per-rank sub-problems sharing scalars only; LORADS_SEPARABLE=0: the general form)
    if separable is None:
        separable = world > 1 and os.environ.get("LORADS_SEPARABLE", "1") != "0"
    s.prepare(world, rank, separable=separable)
    s.attach_hip()
    return s


def golden_trace(name):
    return dict(np.load(os.path.join(GOLD, name + ".trace.npz")))


def golden_solves():
    with open(os.path.join(GOLD, "solve.json")) as f:
        return json.load(f)


def instance_path(name):
    return os.path.join(GOLD, name + ".dat-s")


def linesearch_tau(coef):
    """Host scalar code of the line search (lorads_alm.c:173-227) through the C host."""
    lib = load_oracle()
    k = (C.c_double * 4)(*coef)
    tau = C.c_double(0.0)
    lib.lrd_linesearch_tau.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    n = lib.lrd_linesearch_tau(k, C.byref(tau))
    return tau.value, n


class Mismatch(AssertionError):
    pass


def _cmp(name, got, want, rtol, atol, log):
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    if got.shape != want.shape:
        raise Mismatch("%s: shape %s vs %s" % (name, got.shape, want.shape))
    scale = max(np.max(np.abs(want)), 1e-300) if want.size else 1.0
    err = np.max(np.abs(got - want)) / scale if want.size else 0.0
    log.append((name, err))
    if not np.all(np.abs(got - want) <= atol + rtol * scale):
        raise Mismatch("%s: max rel-to-scale error %.3e (rtol %.1e)" % (name, err, rtol))


def replay_trace(sess, g, rtol=1e-9, atol=1e-13, resync=True):
    """Replays ref_driver's `trace` script (oracle/ref_driver.c: mode_trace) on sess.be and compares
    every output with the reference's.  With resync=True the state (R, lambda, ...) is reset to the
    reference's after each comparison so that errors do not compound across steps.
    Returns the list of (record, error)."""
    be = sess.be
    nb = sess.nblk
    log = []
    rho = float(g["trace_rho"][0])
    n_alm, n_admm = int(g["_nALM"][0]), int(g["_nADMM"][0])
    shapes = [sess.block_shape(k) for k in range(nb)]

    def mat(name, it, k):
        n, r = shapes[k]
        return g["%s_%d_%d" % (name, it, k)].reshape(r, n).T

    for k in range(nb):
        _cmp("R0_%d" % k, be.get_mat(host.MAT_R, k), mat("R", 0, k), 0, 0, log)
    be.init_constr(host.PAIR_RR)
    _cmp("csum_init", be.get_vec(host.VEC_CONSTR_SUM), g["csum_init"], rtol, atol, log)
    lag = be.alm_cal_grad(rho)
    for k in range(nb):
        _cmp("Grad_0_%d" % k, be.get_mat(host.MAT_GRAD, k), mat("Grad", 0, k), rtol, atol, log)
    _cmp("lagsq_0", lag, g["lagsq_0_0"], rtol, atol, log)
    _cmp("pObj_init", be.cal_obj(host.PAIR_RR), g["pObj_init"], rtol, atol, log)
    for it in range(n_alm):
        be.lbfgs_direction(it)
        for k in range(nb):
            _cmp("D_%d_%d" % (it, k), be.get_mat(host.MAT_U, k), mat("D", it, k), rtol * 100, atol, log)
            if resync:
                be.set_mat(host.MAT_U, k, mat("D", it, k))
        p1, p2 = be.alm_q12p12()
        _cmp("q1_%d" % it, be.get_vec(host.VEC_Q1), g["q1_%d_0" % it], rtol, atol, log)
        _cmp("q2_%d" % it, be.get_vec(host.VEC_Q2), g["q2_%d_0" % it], rtol, atol, log)
        _cmp("p12_%d" % it, [p1, p2], g["p12_%d_0" % it], rtol, atol, log)
        coef = be.alm_linesearch_coeffs(rho, p1, p2)
        tau, nroot = linesearch_tau(coef)
        _cmp("tau_%d" % it, [tau, nroot], g["tau_%d_0" % it], rtol * 1e3, atol, log)
        if resync:
            tau = float(g["tau_%d_0" % it][0])
        be.set_y_as_neg_grad()
        be.alm_update_var(tau)
        _cmp("csum_inc_%d" % it, be.get_vec(host.VEC_CONSTR_SUM), g["csum_inc_%d_0" % it], rtol * 10, atol, log)
        lag = be.alm_cal_grad(rho)
        be.set_lbfgs_his_two(tau)
        err1 = be.update_dimacs(host.PAIR_RR)
        for k in range(nb):
            _cmp("R_%d_%d" % (it + 1, k), be.get_mat(host.MAT_R, k), mat("R", it + 1, k), rtol * 10, atol, log)
            _cmp("Grad_%d_%d" % (it + 1, k), be.get_mat(host.MAT_GRAD, k), mat("Grad", it + 1, k), rtol * 100, atol, log)
        _cmp("lagsq_%d" % (it + 1), lag, g["lagsq_%d_0" % (it + 1)], rtol * 100, atol, log)
        _cmp("csum_%d" % it, be.get_vec(host.VEC_CONSTR_SUM), g["csum_%d_0" % it], rtol * 10, atol, log)
        _cmp("err1_%d" % it, err1, g["err1_%d_0" % it], rtol * 100, atol, log)
    be.update_dual_var(rho)
    _cmp("lambda_alm", be.get_vec(host.VEC_LAMBDA), g["lambda_alm"], rtol * 100, atol, log)
    _cmp("pObj_alm", be.cal_obj(host.PAIR_RR), g["pObj_alm"], rtol * 100, atol, log)
    _cmp("dObj_alm", be.cal_dual_obj(), g["dObj_alm"], rtol * 100, atol, log)
    # ---- ADMM part: starts from the reference's own phase-1 warm start (inputs R_warm, lambda_warm)
    rank_warm = [int(x) for x in g["rank_warm"]]
    if rank_warm != [sh[1] for sh in shapes]:
        be.resize_rank(rank_warm)  # the reference's phase 1 grew the rank (AUG_RANK)
        shapes[:] = [(sh[0], r) for sh, r in zip(shapes, rank_warm)]
    for k in range(nb):
        be.set_mat(host.MAT_R, k, mat("R_warm", 0, k))
    be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
    rho = float(g["admm_rho"][0])
    be.alm_to_admm()
    be.init_constr(host.PAIR_UV)
    pobj = be.cal_obj(host.PAIR_UV)
    dobj = be.cal_dual_obj()
    l1 = be.update_dimacs(host.PAIR_UV)
    _cmp("admm_csum_init", be.get_vec(host.VEC_CONSTR_SUM), g["admm_csum_init"], rtol * 100, atol, log)
    _cmp("admm_init_scalars", [pobj, dobj, l1], g["admm_init_scalars"], rtol * 100, atol, log)
    if resync:
        l1 = float(g["admm_init_scalars"][2])
    cg_total = 0
    # The CG solves amplify rounding by the conditioning of (I + A_V^* A_V).  With the state re-synchronised after every step the
    # device's iterates sit 1e-15 ... 3e-14 of scale from the reference's on all twelve golden instances (sparse and dense mode
    # alike; measured round 3, profiles/r03_trace_errors.txt): the bounds are 1e-9 for factors and m-vectors, 1e-11 for the
    # objectives -- four to five orders above what is achieved, four below what they were (1e-5).  Without re-synchronisation
    # (errors compound over the steps) the older bound stays.
    cg_rtol = 1e-9 if resync else max(rtol * 1e4, 1e-7)
    obj_rtol = 1e-11 if resync else cg_rtol
    for it in range(n_admm):
        tol = min(l1 * 1e-2, 1e-8)
        cg_total += be.admm_update_var(rho, tol, 800)
        for k in range(nb):
            _cmp("U_%d_%d" % (it, k), be.get_mat(host.MAT_U, k), mat("U", it, k), cg_rtol, atol, log)
            _cmp("V_%d_%d" % (it, k), be.get_mat(host.MAT_V, k), mat("V", it, k), cg_rtol, atol, log)
        _cmp("csum_uv_%d" % it, be.get_vec(host.VEC_CONSTR_SUM), g["csum_uv_%d_0" % it], cg_rtol, atol, log)
        pobj = be.cal_obj(host.PAIR_UV)
        dobj = be.cal_dual_obj()
        l1 = be.update_dimacs(host.PAIR_UV)
        sc = g["admm_scalars_%d_0" % it]
        _cmp("admm_pobj_%d" % it, pobj, sc[0], obj_rtol, atol, log)
        _cmp("admm_dobj_%d" % it, dobj, sc[1], obj_rtol, atol, log)
        _cmp("admm_err1_%d" % it, l1, sc[2], cg_rtol * 10, 1e-12, log)
        # sparse-mode runs (no dense scratch matrix anywhere: the reference sums in the same sparse order): the CG must stop at
        # the same iteration as the reference's; the reference's dense branch rounds differently, so a test that sits on the
        # threshold may fall the other way there
        dense_mode = bool(np.any(np.asarray(g.get("wsum_is_dense", [0.0])) > 0))
        slack = max(3, 0.02 * sc[4]) if dense_mode else 0
        if abs(cg_total - sc[4]) > slack:
            raise Mismatch("cg iterations %d vs reference %d at ADMM it %d" % (cg_total, sc[4], it))
        log.append(("cg_iters_%d" % it, abs(cg_total - sc[4])))
        _cmp("csum_rr_%d" % it, be.get_vec(host.VEC_CONSTR_SUM), g["csum_rr_%d_0" % it], cg_rtol, atol, log)
        be.update_dual_var(rho)
        _cmp("lambda_%d" % it, be.get_vec(host.VEC_LAMBDA), g["lambda_%d_0" % it], cg_rtol, atol, log)
        if resync:
            for k in range(nb):
                be.set_mat(host.MAT_U, k, mat("U", it, k))
                be.set_mat(host.MAT_V, k, mat("V", it, k))
            be.set_vec(host.VEC_LAMBDA, g["lambda_%d_0" % it])
            be.set_vec(host.VEC_CONSTR_SUM, g["csum_rr_%d_0" % it])
            l1 = float(sc[2])
            cg_total = int(sc[4])
    return log


def trace_worst(log):
    """worst rel-to-scale error of a replay, by group: phase 1, ADMM factors, ADMM m-vectors, ADMM objectives, err1"""
    def worst(pred):
        v = [e for n_, e in log if pred(n_)]
        return max(v) if v else 0.0
    adm = ("U_", "V_", "csum_uv", "admm_", "csum_rr", "lambda_", "cg_")
    return dict(phase1=worst(lambda n_: not n_.startswith(adm) or n_ == "lambda_alm"),
                factors=worst(lambda n_: n_.startswith(("U_", "V_"))),
                vectors=worst(lambda n_: n_.startswith(("csum_uv", "csum_rr", "lambda_")) and n_ != "lambda_alm"),
                objectives=worst(lambda n_: n_.startswith(("admm_pobj", "admm_dobj"))),
                err1=worst(lambda n_: n_.startswith("admm_err1")))


def slack_matrices(prob, lam):
    """S_k = C_k - sum_i lam_i A_ik per cone as scipy CSR, straight from the generator's SDPA entries
    (C = -F0, the sign the reference stores; data/lorads_solver.c:1027-1029)."""
    import scipy.sparse as sp
    out = []
    for k, n in enumerate(prob["blocks"]):
        n = abs(n)  # a negative dimension marks the LP block (diagonal)
        rows, cols, vals = [], [], []
        for mat, blk, i, j, v in prob["entries"]:
            if blk - 1 != k:
                continue
            w = -v if mat == 0 else -lam[mat - 1] * v
            rows.append(i - 1), cols.append(j - 1), vals.append(w)
            if i != j:
                rows.append(j - 1), cols.append(i - 1), vals.append(w)
        out.append(sp.csr_matrix((vals, (rows, cols)), shape=(n, n)))
    return out


def c_norm1(prob):
    """sum of |C_ij| over the full symmetric matrix (LORADSNrm1Obj, data/lorads_solver.c)"""
    t = 0.0
    acc = {}
    for mat, blk, i, j, v in prob["entries"]:
        if mat == 0:
            key = (blk, max(i, j), min(i, j))
            acc[key] = acc.get(key, 0.0) + v
    for (blk, i, j), v in acc.items():
        t += abs(v) * (1 if i == j else 2)
    return t


def exact_dual_infeasibility(prob, lam):
    """(sum, per-block lambda_min) of what calculate_dual_infeasibility_solver accumulates (data/lorads_solver.c:1015-1033):
    |min(lambda_min(S_k), 0)| per SDP cone, and for the LP block every column on its own: sum_i |min(S_ii, 0)|."""
    tot, mins = 0.0, []
    for S, n in zip(slack_matrices(prob, lam), prob["blocks"]):
        if n < 0:
            d = S.diagonal()
            tot += float(np.abs(np.minimum(d, 0.0)).sum())
            mins.append(0.0)
        else:
            e = float(np.linalg.eigvalsh(S.toarray())[0])
            tot += abs(min(e, 0.0))
            mins.append(e)
    return tot, mins


def read_dump(path):
    """records written by oracle/ref_driver.c: [int32 len][name][int64 n][n doubles]"""
    import struct
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from("<i", data, pos)
        pos += 4
        name = data[pos:pos + ln].decode()
        pos += ln
        (n,) = struct.unpack_from("<q", data, pos)
        pos += 8
        out[name] = np.frombuffer(data, dtype="<f8", count=n, offset=pos).copy()
        pos += 8 * n
    return out
