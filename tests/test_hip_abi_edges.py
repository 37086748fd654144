"""Edge cases straight at the C ABI (include/lorads_hip.h) with hand-built tiny problems and numpy as the checker:
malformed input is refused with a message, empty pieces (a cone without constraints, a constraint without entries,
no objective, no constraints at all, a 1 x 1 cone) are handled."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ip, _dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)


class Block(C.Structure):
    _fields_ = [("n", C.c_int32), ("rank", C.c_int32), ("nrow", C.c_int32), ("row_idx", _ip), ("a_ptr", _ip), ("a_row", _ip),
                ("a_col", _ip), ("a_val", _dp), ("c_nnz", C.c_int32), ("c_row", _ip), ("c_col", _ip), ("c_val", _dp),
                ("is_lp", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("m", C.c_int32), ("b", _dp), ("b_nrm1", C.c_double), ("nblocks", C.c_int32), ("blocks", C.POINTER(Block)),
                ("lbfgs_len", C.c_int32), ("device", C.c_int32)]


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


@pytest.fixture(scope="module")
def lib(built):
    lib = C.CDLL(os.path.join(ROOT, "lorads_amd", "lib", "liblorads_hip.so"))
    lib.lorads_hip_last_error.restype = C.c_char_p
    lib.lorads_hip_create.argtypes = [C.POINTER(Problem), C.POINTER(C.c_void_p)]
    lib.lorads_hip_destroy.argtypes = [C.c_void_p]
    lib.lorads_hip_set_mat.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp]
    lib.lorads_hip_get_mat.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp]
    lib.lorads_hip_set_vec.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.lorads_hip_get_vec.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.lorads_hip_init_constr.argtypes = [C.c_void_p, C.c_int32]
    lib.lorads_hip_alm_cal_grad.argtypes = [C.c_void_p, C.c_double, _dp]
    lib.lorads_hip_cal_obj.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.lorads_hip_update_dimacs.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.lorads_hip_alm_to_admm.argtypes = [C.c_void_p]
    lib.lorads_hip_admm_update_var.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32, _ip]
    lib.lorads_hip_resize_rank.argtypes = [C.c_void_p, _ip]
    return lib


class Ctx:
    """cones = [(n, r, rows, A (list of [(i, j, v)] per row, i >= j), Ctriplets)]"""

    def __init__(self, lib, m, b, cones):
        self.lib, self.keep, self.cones, self.m = lib, [], cones, m
        blocks = (Block * max(len(cones), 1))()
        for k, (n, r, rows, A, Ct) in enumerate(cones):
            ptr = np.cumsum([0] + [len(a) for a in A])
            flat = [t for a in A for t in a]
            arrs = [_i(rows), _i(ptr), _i([t[0] for t in flat]), _i([t[1] for t in flat]), _d([t[2] for t in flat]),
                    _i([t[0] for t in Ct]), _i([t[1] for t in Ct]), _d([t[2] for t in Ct])]
            self.keep.append(arrs)
            bk = blocks[k]
            bk.n, bk.rank, bk.nrow = n, r, len(rows)
            bk.row_idx, bk.a_ptr, bk.a_row, bk.a_col, bk.a_val = arrs[0][1], arrs[1][1], arrs[2][1], arrs[3][1], arrs[4][1]
            bk.c_nnz, bk.c_row, bk.c_col, bk.c_val = len(Ct), arrs[5][1], arrs[6][1], arrs[7][1]
        bb = _d(b if m else [0.0])
        self.keep.append((bb, blocks))
        self.prob = Problem(m, bb[1], float(np.abs(np.asarray(b)).sum()) if m else 0.0, len(cones), blocks, 2, -1)
        self.h = C.c_void_p()
        self.rc = lib.lorads_hip_create(C.byref(self.prob), C.byref(self.h))

    def close(self):
        if self.h:
            self.lib.lorads_hip_destroy(self.h)
            self.h = C.c_void_p()

    def dense(self, k):
        n, r, rows, A, Ct = self.cones[k]
        Cm = np.zeros((n, n))
        for i, j, v in Ct:
            Cm[i, j] += v
            if i != j:
                Cm[j, i] += v
        As = []
        for a in A:
            M = np.zeros((n, n))
            for i, j, v in a:
                M[i, j] += v
                if i != j:
                    M[j, i] += v
            As.append(M)
        return Cm, As


def test_malformed_input_is_refused(lib):
    up = Ctx(lib, 1, [1.0], [(3, 2, [0], [[(0, 2, 1.0)]], [])])  # row < col: not lower-triangular
    assert up.rc != 0 and b"lower-triangular" in lib.lorads_hip_last_error()
    oob = Ctx(lib, 1, [1.0], [(3, 2, [5], [[(1, 1, 1.0)]], [])])  # constraint index >= m
    assert oob.rc != 0 and b"constraint index" in lib.lorads_hip_last_error()
    big = Ctx(lib, 1, [1.0], [(3, 2, [0], [[(3, 0, 1.0)]], [])])  # row >= n
    assert big.rc != 0
    rk = Ctx(lib, 1, [1.0], [(3, 600, [0], [[(1, 1, 1.0)]], [])])  # rank beyond the row kernels
    assert rk.rc != 0 and b"rank" in lib.lorads_hip_last_error()


def test_empty_pieces_and_tiny_cones(lib):
    """cone 0: 4 x 4 with one real constraint, one constraint WITHOUT entries and an objective; cone 1: 1 x 1 with a
    constraint and no objective; cone 2: 3 x 3 without any constraint.  Gradient, objective, residual vs numpy."""
    m, b = 3, [1.0, 0.25, 2.0]
    cones = [(4, 2, [0, 1], [[(0, 0, 1.0), (2, 1, 0.5)], []], [(1, 0, -1.0), (3, 3, 2.0)]),
             (1, 1, [2], [[(0, 0, 3.0)]], []),
             (3, 2, [], [], [(2, 0, 0.5), (1, 1, 1.0)])]
    cx = Ctx(lib, m, b, cones)
    try:
        assert cx.rc == 0, lib.lorads_hip_last_error()
        rng = np.random.default_rng(3)
        Rs = [rng.standard_normal((n, r)) for n, r, *_ in cones]
        lam = rng.standard_normal(m)
        for k, R in enumerate(Rs):
            f = np.asfortranarray(R)
            assert lib.lorads_hip_set_mat(cx.h, 0, k, f.ctypes.data_as(_dp)) == 0
        assert lib.lorads_hip_set_vec(cx.h, 0, _d(lam)[1]) == 0
        assert lib.lorads_hip_init_constr(cx.h, 0) == 0
        csum = np.zeros(m)
        assert lib.lorads_hip_get_vec(cx.h, 1, csum.ctypes.data_as(_dp)) == 0
        want = np.zeros(m)
        obj = 0.0
        for k, (n, r, rows, A, Ct) in enumerate(cones):
            Cm, As = cx.dense(k)
            X = Rs[k] @ Rs[k].T
            obj += float((Cm * X).sum())
            for i, Ai in zip(rows, As):
                want[i] += float((Ai * X).sum())
        assert np.allclose(csum, want, rtol=1e-13, atol=1e-14)
        rho, lag = 0.8, C.c_double()
        assert lib.lorads_hip_alm_cal_grad(cx.h, rho, C.byref(lag)) == 0
        M1 = -lam - rho * np.asarray(b) + rho * want
        tot = 0.0
        for k, (n, r, rows, A, Ct) in enumerate(cones):
            Cm, As = cx.dense(k)
            S = Cm + sum(M1[i] * Ai for i, Ai in zip(rows, As)) if rows else Cm
            G = 2 * S @ Rs[k]
            got = np.zeros((n, r), order="F")
            assert lib.lorads_hip_get_mat(cx.h, 3, k, got.ctypes.data_as(_dp)) == 0
            assert np.allclose(got, G, rtol=1e-12, atol=1e-13), k
            tot += float((G * G).sum())
        assert lag.value == pytest.approx(tot, rel=1e-12)
        po, e1 = C.c_double(), C.c_double()
        assert lib.lorads_hip_cal_obj(cx.h, 0, C.byref(po)) == 0 and po.value == pytest.approx(obj, rel=1e-12)
        assert lib.lorads_hip_update_dimacs(cx.h, 0, C.byref(e1)) == 0
        assert e1.value == pytest.approx(np.linalg.norm(np.asarray(b) - want) / (1 + np.abs(b).sum()), rel=1e-12)
        # one ADMM sweep runs through all three cones (the constraint-free cone has a trivial operator)
        assert lib.lorads_hip_alm_to_admm(cx.h) == 0 and lib.lorads_hip_init_constr(cx.h, 1) == 0
        its = C.c_int32()
        assert lib.lorads_hip_admm_update_var(cx.h, 1.0, 1e-10, 100, C.byref(its)) == 0
        for k, (n, r, *_rest) in enumerate(cones):
            U = np.zeros((n, r), order="F")
            assert lib.lorads_hip_get_mat(cx.h, 1, k, U.ctypes.data_as(_dp)) == 0
            assert np.all(np.isfinite(U))
        # rank growth keeps the old columns (AUG_RANK)
        newr = _i([3, 2, 4])
        assert lib.lorads_hip_resize_rank(cx.h, newr[1]) == 0
        R0 = np.zeros((4, 3), order="F")
        assert lib.lorads_hip_get_mat(cx.h, 0, 0, R0.ctypes.data_as(_dp)) == 0
        assert np.array_equal(R0[:, :2], Rs[0]) and R0[0, 2] == 1.0  # one new column: 1/sqrt(1) on its leading diagonal
    finally:
        cx.close()


def test_problem_without_constraints(lib):
    cx = Ctx(lib, 0, [], [(5, 2, [], [], [(0, 0, 1.0), (4, 2, -0.5)])])
    try:
        assert cx.rc == 0, lib.lorads_hip_last_error()
        R = np.asfortranarray(np.random.default_rng(1).standard_normal((5, 2)))
        assert lib.lorads_hip_set_mat(cx.h, 0, 0, R.ctypes.data_as(_dp)) == 0
        assert lib.lorads_hip_init_constr(cx.h, 0) == 0
        lag, e1 = C.c_double(), C.c_double()
        assert lib.lorads_hip_alm_cal_grad(cx.h, 1.0, C.byref(lag)) == 0
        Cm, _ = cx.dense(0)
        assert lag.value == pytest.approx(float(((2 * Cm @ R) ** 2).sum()), rel=1e-12)
        assert lib.lorads_hip_update_dimacs(cx.h, 0, C.byref(e1)) == 0 and e1.value == 0.0
    finally:
        cx.close()
