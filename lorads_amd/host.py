"""ctypes mirror of the plain-C host (lorads_amd/csrc/host) and of its operator table.

The table `Backend` is field-for-field `lrd_backend` (csrc/host/lorads_host.h), which itself is
slot-for-slot the reference's `lorads_func` (src_semi/data/def_lorads_solver.h:109-127) plus the
non-table calls on the path.  Tests read like the reference's own call sites:

    be.init_constr(PAIR_RR); lag = be.alm_cal_grad(rho); be.lbfgs_direction(it); ...

The product attaches ONLY the HIP table (`Session.attach_hip`), which raises if the HIP library is
missing -- there is no CPU fallback here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
DEV_LIB = os.path.join(LIB_DIR, "liblorads_hip_dev.so")   # development build: the product library + lorads_hip_ubench (csrc/Makefile `dev`)

PAIR_RR, PAIR_UV = 0, 1
MAT_R, MAT_U, MAT_V, MAT_GRAD = 0, 1, 2, 3
VEC_LAMBDA, VEC_CONSTR_SUM, VEC_Q1, VEC_Q2 = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)


class BackendStruct(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("name", C.c_char_p),
        ("init_constr", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)),
        ("alm_cal_grad", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, _dp)),
        ("lbfgs_direction", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)),
        ("alm_q12p12", C.CFUNCTYPE(C.c_int, C.c_void_p, _dp)),
        ("alm_linesearch_coeffs", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, _dp)),
        ("set_y_as_neg_grad", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("alm_update_var", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)),
        ("set_lbfgs_his_two", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)),
        ("update_dimacs", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _dp)),
        ("cal_obj", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _dp)),
        ("admm_update_var", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, _ip)),
        ("update_dual_var", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)),
        ("cal_dual_obj", C.CFUNCTYPE(C.c_int, C.c_void_p, _dp)),
        ("alm_to_admm", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("average_uv_to_v", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("scale_obj", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)),
        ("resize_rank", C.CFUNCTYPE(C.c_int, C.c_void_p, _ip)),
        ("set_mat", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, _dp)),
        ("get_mat", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, _dp)),
        ("set_vec", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _dp)),
        ("get_vec", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _dp)),
        ("set_allreduce", C.CFUNCTYPE(C.c_int, C.c_void_p, ALLREDUCE_FN, C.c_void_p)),
        ("destroy", C.CFUNCTYPE(None, C.c_void_p)),
        ("admm_step", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, _dp)),
        ("dual_infeasibility", C.CFUNCTYPE(C.c_int, C.c_void_p, _dp)),
        ("alm_front", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_int, _dp)),
        ("alm_step", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, _dp)),
    ]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


class Backend:
    """Pythonic view of one operator table (the one a Session owns after attach)."""

    def __init__(self, struct_ptr, session):
        self._s = struct_ptr.contents
        self._session = session

    @property
    def name(self):
        return self._s.name.decode()

    def init_constr(self, pair):
        _check(self._s.init_constr(self._s.ctx, pair), "init_constr")

    def alm_cal_grad(self, rho):
        v = C.c_double()
        _check(self._s.alm_cal_grad(self._s.ctx, rho, C.byref(v)), "alm_cal_grad")
        return v.value

    def lbfgs_direction(self, inner_iter):
        _check(self._s.lbfgs_direction(self._s.ctx, inner_iter), "lbfgs_direction")

    def alm_q12p12(self):
        p = (C.c_double * 2)()
        _check(self._s.alm_q12p12(self._s.ctx, p), "alm_q12p12")
        return p[0], p[1]

    def alm_linesearch_coeffs(self, rho, p1, p2):
        k = (C.c_double * 4)()
        _check(self._s.alm_linesearch_coeffs(self._s.ctx, rho, p1, p2, k), "alm_linesearch_coeffs")
        return [k[i] for i in range(4)]

    def set_y_as_neg_grad(self):
        _check(self._s.set_y_as_neg_grad(self._s.ctx), "set_y_as_neg_grad")

    def alm_update_var(self, tau):
        _check(self._s.alm_update_var(self._s.ctx, tau), "alm_update_var")

    def set_lbfgs_his_two(self, tau):
        _check(self._s.set_lbfgs_his_two(self._s.ctx, tau), "set_lbfgs_his_two")

    def update_dimacs(self, pair):
        v = C.c_double()
        _check(self._s.update_dimacs(self._s.ctx, pair, C.byref(v)), "update_dimacs")
        return v.value

    def cal_obj(self, pair):
        v = C.c_double()
        _check(self._s.cal_obj(self._s.ctx, pair, C.byref(v)), "cal_obj")
        return v.value

    def admm_update_var(self, rho, cg_tol, cg_max_iter=800):
        it = C.c_int()
        _check(self._s.admm_update_var(self._s.ctx, rho, cg_tol, cg_max_iter, C.byref(it)), "admm_update_var")
        return it.value

    @property
    def has_admm_step(self):
        return bool(self._s.admm_step)

    def admm_step(self, rho, cg_tol, cg_max_iter=800):
        """fused ADMM iteration (optional slot): returns (cg_iters, pobj, dobj, err1)"""
        o = (C.c_double * 4)()
        _check(self._s.admm_step(self._s.ctx, rho, cg_tol, cg_max_iter, o), "admm_step")
        return int(o[0]), o[1], o[2], o[3]

    def update_dual_var(self, rho):
        _check(self._s.update_dual_var(self._s.ctx, rho), "update_dual_var")

    def cal_dual_obj(self):
        v = C.c_double()
        _check(self._s.cal_dual_obj(self._s.ctx, C.byref(v)), "cal_dual_obj")
        return v.value

    @property
    def has_alm_step(self):
        return bool(self._s.alm_step) and bool(self._s.alm_front)

    def alm_front(self, rho, inner):
        """(p1, p2, [a, b, c, d]) of the direction for inner-iteration counter `inner` (optional slot)"""
        o = (C.c_double * 6)()
        _check(self._s.alm_front(self._s.ctx, rho, inner, o), "alm_front")
        return o[0], o[1], [o[2], o[3], o[4], o[5]]

    def alm_step(self, rho, tau, next_inner):
        """finish the inner iteration with step tau and pre-compute the next direction (optional slot):
        (lagNormSq, err1, p1, p2, [a, b, c, d])"""
        o = (C.c_double * 8)()
        _check(self._s.alm_step(self._s.ctx, rho, tau, next_inner, o), "alm_step")
        return o[0], o[1], o[2], o[3], [o[4], o[5], o[6], o[7]]

    @property
    def has_dual_infeasibility(self):
        return bool(self._s.dual_infeasibility)

    def dual_infeasibility(self):
        """sum over the table's cones of |min(lambda_min(C_k - A_k^*(lambda)), 0)| (optional slot)"""
        v = C.c_double()
        _check(self._s.dual_infeasibility(self._s.ctx, C.byref(v)), "dual_infeasibility")
        return v.value

    def alm_to_admm(self):
        _check(self._s.alm_to_admm(self._s.ctx), "alm_to_admm")

    def average_uv_to_v(self):
        _check(self._s.average_uv_to_v(self._s.ctx), "average_uv_to_v")

    def scale_obj(self, s):
        _check(self._s.scale_obj(self._s.ctx, s), "scale_obj")

    def resize_rank(self, new_rank):
        arr = (C.c_int * len(new_rank))(*[int(x) for x in new_rank])
        _check(self._s.resize_rank(self._s.ctx, arr), "resize_rank")
        self._session._rank_override = [int(x) for x in new_rank]

    def set_mat(self, which, blk, a):
        """a: (n, r) array, any layout; sent column-major like the reference's matElem."""
        a = np.asfortranarray(a, dtype=np.float64)
        _check(self._s.set_mat(self._s.ctx, which, blk, a.ctypes.data_as(_dp)), "set_mat")

    def get_mat(self, which, blk):
        n, r = self._session.block_shape(blk)
        out = np.empty((n, r), dtype=np.float64, order="F")
        _check(self._s.get_mat(self._s.ctx, which, blk, out.ctypes.data_as(_dp)), "get_mat")
        return out

    def set_vec(self, which, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        _check(self._s.set_vec(self._s.ctx, which, v.ctypes.data_as(_dp)), "set_vec")

    def get_vec(self, which):
        out = np.empty(self._session.m, dtype=np.float64)
        _check(self._s.get_vec(self._s.ctx, which, out.ctypes.data_as(_dp)), "get_vec")
        return out


def _bind(lib):
    lib.lrd_session_open.restype = C.c_void_p
    lib.lrd_session_open.argtypes = [C.c_char_p]
    lib.lrd_session_from_triplets.restype = C.c_void_p
    lib.lrd_session_from_triplets.argtypes = [C.c_int, _dp, C.c_int, _ip, C.c_int64, _ip, _ip, _ip, _ip, _dp]
    lib.lrd_session_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.lrd_session_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lrd_session_problem.restype = C.c_void_p
    lib.lrd_session_problem.argtypes = [C.c_void_p]
    lib.lrd_session_backend.restype = C.POINTER(BackendStruct)
    lib.lrd_session_backend.argtypes = [C.c_void_p]
    lib.lrd_session_attach.argtypes = [C.c_void_p, C.POINTER(BackendStruct)]
    lib.lrd_session_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    lib.lrd_session_solve.argtypes = [C.c_void_p]
    lib.lrd_session_alm.argtypes = [C.c_void_p]
    lib.lrd_session_alm_to_admm.argtypes = [C.c_void_p]
    lib.lrd_session_alm_to_admm.restype = None
    lib.lrd_session_admm.argtypes = [C.c_void_p, C.c_int]
    lib.lrd_session_results.argtypes = [C.c_void_p, _dp]
    lib.lrd_session_results2.argtypes = [C.c_void_p, _dp]
    lib.lrd_session_dual_infeasibility.argtypes = [C.c_void_p, _dp]
    lib.lrd_session_block_info.argtypes = [C.c_void_p, C.c_int] + [_ip] * 8
    lib.lrd_session_dims.argtypes = [C.c_void_p, _ip, _ip, _ip]
    lib.lrd_session_start.restype = _dp
    lib.lrd_session_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lrd_session_current_rank.argtypes = [C.c_void_p, C.c_int]
    lib.lrd_session_close.argtypes = [C.c_void_p]
    lib.lrd_session_close.restype = None
    lib.lrd_session_params_ptr = None
    return lib


_host_lib = None


def host_lib():
    """liblorads_host.so: the product's plain-C host, built by __graft_entry__.build()."""
    global _host_lib
    if _host_lib is None:
        path = os.path.join(LIB_DIR, "liblorads_host.so")
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python __graft_entry__.py` (build()) first" % path)
        _host_lib = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL))
        _host_lib.lrd_hip_backend_create.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(BackendStruct)]
    return _host_lib


RESULT_KEYS = ["pObj", "dObj", "constrVio1", "pdGap", "alm_outer", "alm_inner", "alm_rho", "admm_iter", "cg_iter",
               "admm_rho", "t_alm", "t_admm", "status", "admm_iters_first", "cg_iters_first", "constrVioInf"]


class Session:
    """One problem + parameter block + start point (+ solver once a table is attached)."""

    def __init__(self, lib, handle):
        if not handle:
            raise RuntimeError("could not open the problem")
        self.lib = lib
        self.h = C.c_void_p(handle)
        self.be = None
        self._rank_override = None
        self._keep = []

    @classmethod
    def open(cls, fname, lib=None):
        lib = lib or host_lib()
        return cls(lib, lib.lrd_session_open(os.fsencode(fname)))

    @classmethod
    def from_triplets(cls, m, b, dims, mat, blk, row, col, val, lib=None):
        lib = lib or host_lib()
        b = np.ascontiguousarray(b, dtype=np.float64)
        dims = np.ascontiguousarray(dims, dtype=np.int32)
        arrs = [np.ascontiguousarray(x, dtype=np.int32) for x in (mat, blk, row, col)]
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = lib.lrd_session_from_triplets(int(m), b.ctypes.data_as(_dp), len(dims), dims.ctypes.data_as(_ip),
                                          len(val), *[a.ctypes.data_as(_ip) for a in arrs], val.ctypes.data_as(_dp))
        return cls(lib, h)

    def set_params(self, **kw):
        for k, v in kw.items():
            if isinstance(v, (bool, np.bool_)) or (isinstance(v, (int, np.integer)) and not isinstance(v, bool)):
                txt = str(int(v))
            elif isinstance(v, (float, np.floating)):
                txt = repr(float(v))  # repr(np.float64(x)) is not a number literal
            else:
                txt = str(v)
            if self.lib.lrd_session_set_param(self.h, k.encode(), txt.encode()):
                raise KeyError("unknown parameter %s" % k)

    def prepare(self, world=1, rank=0, separable=False):
        """separable=True (HIP backend only): when no constraint touches cones of two ranks, this process keeps the sub-problem
        over its own constraints (m, b, row indices local) and the library shares only scalars with the other ranks
        (lorads_hip_set_separable); self.separable tells whether the deal was, self.constraint_map[i] = index of local
        constraint i in the file, self.m_global = constraints of the file."""
        self.lib.lrd_session_prepare_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _check(self.lib.lrd_session_prepare_sharded(self.h, world, rank, int(bool(separable))), "prepare")
        m, nb, nbg = C.c_int(), C.c_int(), C.c_int()
        self.lib.lrd_session_dims(self.h, C.byref(m), C.byref(nb), C.byref(nbg))
        self.m, self.nblk, self.nblk_global = m.value, nb.value, nbg.value
        mg = C.c_int()
        cmap = np.zeros(max(self.m, 1), dtype=np.int32)
        self.lib.lrd_session_separable.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.separable = bool(self.lib.lrd_session_separable(self.h, C.byref(mg), cmap.ctypes.data_as(C.POINTER(C.c_int))))
        self.m_global = mg.value
        self.constraint_map = cmap[:self.m] if self.separable else np.arange(self.m, dtype=np.int32)

    def block_info(self, k):
        v = [C.c_int() for _ in range(8)]
        _check(self.lib.lrd_session_block_info(self.h, k, *[C.byref(x) for x in v]), "block_info")
        keys = ["n", "rank", "nrow", "na", "nc", "np", "dense_mode", "cone_sparse"]
        d = dict(zip(keys, [x.value for x in v]))
        if self._rank_override is not None:
            d["rank"] = self._rank_override[k]
        elif self.be is not None:
            d["rank"] = self.lib.lrd_session_current_rank(self.h, k)
        return d

    def block_shape(self, k):
        d = self.block_info(k)
        return d["n"], d["rank"]

    def start_point(self, which, k):
        d = self.block_info(k)
        p = self.lib.lrd_session_start(self.h, which, k)
        return np.ctypeslib.as_array(p, shape=(d["rank"], d["n"])).T.copy()

    def problem_ptr(self):
        return self.lib.lrd_session_problem(self.h)

    def attach(self, struct):
        _check(self.lib.lrd_session_attach(self.h, C.byref(struct)), "attach")
        self.be = Backend(self.lib.lrd_session_backend(self.h), self)
        return self.be

    def attach_hip(self, lbfgs_len=2, libpath=None):
        """Wire the operator table to the HIP C-ABI library (include/lorads_hip.h).  Raises when the
        library or a GPU is missing: the product has no other backend."""
        st = BackendStruct()
        path = libpath or os.path.join(LIB_DIR, "liblorads_hip.so")
        self._hip_path = path
        rc = self.lib.lrd_hip_backend_create(self.problem_ptr(), lbfgs_len, os.fsencode(path), C.byref(st))
        if rc != 0:
            raise RuntimeError("HIP backend unavailable (code %d): %s -- the product has no CPU fallback" % (rc, path))
        return self.attach(st)

    # ---- measurement hooks of the HIP library (bench.py)
    def _hip(self):
        if getattr(self, "_hiplib", None) is None:
            lib = C.CDLL(getattr(self, "_hip_path", None) or os.path.join(LIB_DIR, "liblorads_hip.so"))
            lib.lorads_hip_profile.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.lorads_hip_profile_read.argtypes = [C.c_void_p, _dp]
            lib.lorads_hip_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
            lib.lorads_hip_sync.argtypes = [C.c_void_p]
            lib.lorads_hip_operator_kind.argtypes = [C.c_void_p, C.c_int, _ip]
            lib.lorads_hip_stream.restype = C.c_void_p
            lib.lorads_hip_stream.argtypes = [C.c_void_p]
            lib.lorads_hip_set_allreduce_stream_ordered.argtypes = [C.c_void_p, C.c_int]
            self.lib.lrd_hip_backend_raw_ctx.restype = C.c_void_p
            self.lib.lrd_hip_backend_raw_ctx.argtypes = [C.POINTER(BackendStruct)]
            self._hiplib = lib
            self._hipctx = C.c_void_p(self.lib.lrd_hip_backend_raw_ctx(self.lib.lrd_session_backend(self.h)))
        return self._hiplib, self._hipctx

    def hip_profile(self, enable, sample_every=8):
        lib, ctx = self._hip()
        _check(lib.lorads_hip_profile(ctx, int(enable), int(sample_every)), "profile")

    def hip_profile_target(self, target):
        """0: time CG operator applications (default), 1: time solve fronts"""
        lib, ctx = self._hip()
        lib.lorads_hip_profile_target.argtypes = [C.c_void_p, C.c_int]
        _check(lib.lorads_hip_profile_target(ctx, int(target)), "profile_target")

    def hip_profile_read(self):
        lib, ctx = self._hip()
        out = (C.c_double * 8)()
        _check(lib.lorads_hip_profile_read(ctx, out), "profile_read")
        keys = ["matvec_launches", "speculation_misses", "cg_iters", "cg_solves", "sampled", "sampled_ms", "spmm_sampled",
                "spmm_sampled_ms"]
        return dict(zip(keys, [out[i] for i in range(8)]))

    def hip_profile_samples(self, cap=4096):
        """milliseconds of every operator application timed since hip_profile(1, ...) (drains the event pool)"""
        lib, ctx = self._hip()
        lib.lorads_hip_profile_samples.argtypes = [C.c_void_p, _dp, C.c_int, _ip]
        buf, n = (C.c_double * cap)(), C.c_int()
        _check(lib.lorads_hip_profile_samples(ctx, buf, cap, C.byref(n)), "profile_samples")
        return [buf[i] for i in range(min(cap, n.value))]

    def hip_time_operator(self, reps):
        """milliseconds of `reps` applications of the live CG operator of cone 0, back to back (see lorads_hip_dev.h)"""
        lib, ctx = self._hip()
        ms = C.c_double()
        lib.lorads_hip_time_operator.argtypes = [C.c_void_p, C.c_int, _dp]
        rc = lib.lorads_hip_time_operator(ctx, int(reps), C.byref(ms))
        if rc:
            lib.lorads_hip_last_error.restype = C.c_char_p
            raise RuntimeError("time_operator: %s" % (lib.lorads_hip_last_error() or b"?").decode())
        return ms.value

    def hip_ubench(self, which, reps):
        """milliseconds of `reps` back-to-back launches of kernel variant `which`: DEVELOPMENT build only -- the session must have
        been attached with attach_hip(libpath=host.DEV_LIB) (profiles/tools/ubench.py)"""
        lib, ctx = self._hip()
        if not hasattr(lib, "lorads_hip_ubench"):
            raise RuntimeError("ubench: not in the product library (attach the session to host.DEV_LIB)")
        lib.lorads_hip_ubench.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        ms = C.c_double()
        rc = lib.lorads_hip_ubench(ctx, int(which), int(reps), C.byref(ms))
        if rc:
            lib.lorads_hip_last_error.restype = C.c_char_p
            raise RuntimeError("ubench: %s" % (lib.lorads_hip_last_error() or b"?").decode())
        return ms.value

    def hip_algorithmic_bytes(self, blk=0):
        lib, ctx = self._hip()
        a, b = C.c_double(), C.c_double()
        _check(lib.lorads_hip_algorithmic_bytes(ctx, blk, C.byref(a), C.byref(b)), "algorithmic_bytes")
        return a.value, b.value

    def hip_dual_infeasibility(self, tol=1e-2, ncv=40, max_restarts=600):
        """(sum_k |min(lambda_min_k, 0)|, per-cone lambda_min, S x products) straight from the C ABI"""
        lib, ctx = self._hip()
        lib.lorads_hip_dual_infeasibility.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, _dp, _dp, _ip]
        nb = self.nblk
        v, lm, mv = C.c_double(), (C.c_double * max(nb, 1))(), C.c_int()
        _check(lib.lorads_hip_dual_infeasibility(ctx, tol, ncv, max_restarts, C.byref(v), lm, C.byref(mv)), "dual_infeasibility")
        return v.value, [lm[i] for i in range(nb)], mv.value

    def hip_block_image(self, blk=0):
        """what lorads_hip_create built for cone blk (see lorads_hip_dev.h)"""
        lib, ctx = self._hip()
        out = (C.c_int64 * 16)()
        lib.lorads_hip_block_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_block_image(ctx, blk, out), "block_image")
        keys = ["n", "rank", "nrow", "na", "nc", "pattern_a", "pattern_union", "dense_c", "dense_a", "diag_only", "entry_only", "use_cw",
                "has_gram", "front_cw", "slot_width", "bip_rows0"]
        return dict(zip(keys, [int(out[i]) for i in range(16)]))

    def hip_graph_stats(self):
        """{captured, replayed, held, enabled} of the launch-chain replay (hipGraph) of this context"""
        lib, ctx = self._hip()
        out = (C.c_int64 * 4)()
        lib.lorads_hip_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_graph_stats(ctx, out), "graph_stats")
        return dict(zip(["captured", "replayed", "held", "enabled"], [int(out[i]) for i in range(4)]))

    def hip_persist_stats(self):
        """the one-launch ADMM iteration of Max-Cut-type cones (csrc/hip/persist.inc): {iterations run that way, available now,
        workgroups, rows per lane group, column steps, LDS bytes}"""
        lib, ctx = self._hip()
        out = (C.c_int64 * 6)()
        lib.lorads_hip_persist_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_persist_stats(ctx, out), "persist_stats")
        return dict(zip(["iterations", "available", "workgroups", "rows", "column_steps", "lds_bytes"], [int(out[i]) for i in range(6)]))

    def hip_launch_count(self):
        """kernels this context has enqueued so far"""
        lib, ctx = self._hip()
        out = C.c_int64(0)
        lib.lorads_hip_launch_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_launch_count(ctx, C.byref(out)), "launch_count")
        return int(out.value)

    def hip_lbfgs_team_stats(self):
        """phase 1's one-launch L-BFGS history update + direction (csrc/hip/lbfgs_team.inc): {launches, available, workgroups, pairs}"""
        lib, ctx = self._hip()
        out = (C.c_int64 * 4)()
        lib.lorads_hip_lbfgs_team_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_lbfgs_team_stats(ctx, out), "lbfgs_team_stats")
        return dict(zip(("launches", "available", "workgroups", "pairs"), [int(out[i]) for i in range(4)]))

    def hip_persist_stamps(self, enable=True):
        """100 MHz clock of cone 0's leader workgroup at the phase boundaries of the latest one-launch iteration (0: not taken)"""
        lib, ctx = self._hip()
        out = (C.c_uint64 * 16)()
        lib.lorads_hip_persist_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        _check(lib.lorads_hip_persist_stamps(ctx, 1 if enable else 0, out), "persist_stamps")
        return [int(out[i]) for i in range(16)]

    def hip_presolve_stats(self):
        """{device: patterns built by the device sorts, checked: of these compared with the host construction}"""
        lib, ctx = self._hip()
        out = (C.c_int64 * 2)()
        lib.lorads_hip_presolve_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_presolve_stats(ctx, out), "presolve_stats")
        return {"device": int(out[0]), "checked": int(out[1])}

    def hip_operator_kind(self, blk=0):
        lib, ctx = self._hip()
        k = C.c_int()
        _check(lib.lorads_hip_operator_kind(ctx, blk, C.byref(k)), "operator_kind")
        base = ["k_pairdots+k_sgram+k_spmm2", "k_pairdots+k_cv+k_sval+k_spmm2", "k_op_diag", "k_op_entry", "k_cw+k_spmm_ell"][k.value & 15]
        if k.value & 32:   # bipartite entry graph: one launch per colour
            base = "k_op_entry_bip+k_op_entry_bip"
        return base + ("+k_dense_cx_b(dense A_i)" if k.value & 16 else "")

    def hip_stream(self):
        """hipStream_t of the library as an integer (torch.cuda.ExternalStream takes it)"""
        lib, ctx = self._hip()
        return lib.lorads_hip_stream(ctx)

    def hip_allreduce_stream_ordered(self, on):
        lib, ctx = self._hip()
        _check(lib.lorads_hip_set_allreduce_stream_ordered(ctx, int(on)), "set_allreduce_stream_ordered")

    def hip_selfcheck_allreduce(self):
        lib, ctx = self._hip()
        lib.lorads_hip_selfcheck_allreduce.argtypes = [C.c_void_p]
        _check(lib.lorads_hip_selfcheck_allreduce(ctx), "selfcheck_allreduce")

    def hip_sync(self):
        lib, ctx = self._hip()
        _check(lib.lorads_hip_sync(ctx), "sync")

    def set_allreduce(self, fn):
        """fn(ptr:int, count:int, on_device:bool) -> sums in place over ranks."""
        def _cb(user, buf, count, on_device):
            try:
                fn(buf, count, bool(on_device))
                return 0
            except Exception as e:  # noqa: BLE001 - must not unwind into C
                print("allreduce hook failed:", e)
                return 1
        cb = ALLREDUCE_FN(_cb)
        self._keep.append(cb)
        self.lib.lrd_session_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        _check(self.lib.lrd_session_set_allreduce(self.h, cb, None), "set_allreduce")

    def set_allreduce_native(self, fn_ptr, user):
        """a C function of type lorads_hip_allreduce_fn (address) and its user pointer: no Python in the hook"""
        self.lib.lrd_session_set_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.lib.lrd_session_set_allreduce(self.h, fn_ptr, C.c_void_p(user)), "set_allreduce")

    def set_scalar_exchange_shm(self, name, world, rank):
        """Separable shards on one node: the evaluation's four scalars are summed by the ranks' hosts through a page of POSIX shared
        memory (csrc/host/shmx.c) instead of a collective on the stream (lorads_hip_set_scalar_exchange).  name: '/...', the same on
        every rank and unique per run.  Returns the exchange's handle (closed by close())."""
        lib, ctx = self._hip()
        h = C.c_void_p()
        self.lib.lrd_shmx_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib.lrd_shmx_open(name.encode(), int(world), int(rank), C.byref(h)), "shmx_open")
        lib.lorads_hip_set_scalar_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib.lorads_hip_set_scalar_exchange(ctx, C.cast(self.lib.lrd_shmx_hook, C.c_void_p), h), "set_scalar_exchange")
        self._shmx = h
        return h

    def hip_scalar_exchange_count(self):
        lib, ctx = self._hip()
        n = C.c_int64()
        lib.lorads_hip_scalar_exchange_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(lib.lorads_hip_scalar_exchange_count(ctx, C.byref(n)), "scalar_exchange_count")
        return int(n.value)

    def clear_scalar_exchange(self):
        lib, ctx = self._hip()
        lib.lorads_hip_set_scalar_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib.lorads_hip_set_scalar_exchange(ctx, None, None), "set_scalar_exchange")

    def shmx_allreduce(self, values):
        """sums `values` (<= 16 doubles) over the ranks through the exchange opened by set_scalar_exchange_shm (tests, agreement steps)"""
        v = (C.c_double * len(values))(*values)
        self.lib.lrd_shmx_allreduce.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        _check(self.lib.lrd_shmx_allreduce(self._shmx, v, len(values)), "shmx_allreduce")
        return list(v)

    def use_fused_step(self, on):
        self.lib.lrd_session_use_fused_step.argtypes = [C.c_void_p, C.c_int]
        _check(self.lib.lrd_session_use_fused_step(self.h, int(on)), "use_fused_step")

    def solve(self):
        _check(self.lib.lrd_session_solve(self.h), "solve")
        return self.results()

    def alm(self):
        return self.lib.lrd_session_alm(self.h)

    def alm_to_admm(self):
        self.lib.lrd_session_alm_to_admm(self.h)

    def admm(self, iter_ceiling):
        return self.lib.lrd_session_admm(self.h, iter_ceiling)

    def admm_steps(self, steps, rho, err1):
        """`steps` ADMM iterations in the C host loop (no per-iteration Python); returns (err1, cg, pobj, dobj)"""
        io = (C.c_double * 4)(err1, 0.0, 0.0, 0.0)
        self.lib.lrd_session_admm_steps.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp]
        _check(self.lib.lrd_session_admm_steps(self.h, int(steps), float(rho), io), "admm_steps")
        return io[0], int(io[1]), io[2], io[3]

    def results(self):
        out = (C.c_double * 16)()
        _check(self.lib.lrd_session_results(self.h, out), "results")
        res = dict(zip(RESULT_KEYS, [out[i] for i in range(16)]))
        o2 = (C.c_double * 4)()
        _check(self.lib.lrd_session_results2(self.h, o2), "results2")
        res.update(dual_infeas_l1=o2[0], dual_infeas_inf=o2[1], t_dual_infeas=o2[2], scale_obj_his=o2[3])
        return res

    def dual_infeasibility(self):
        """DIMACS error 2 of the current multipliers, data/lorads_solver.c:1007-1037 (-1: slot missing)"""
        v = C.c_double()
        self.lib.lrd_session_dual_infeasibility(self.h, C.byref(v))
        return v.value

    def close(self):
        if self.h:
            self.lib.lrd_session_close(self.h)
            self.h = None
        if getattr(self, "_shmx", None):
            self.lib.lrd_shmx_close.argtypes = [C.c_void_p]
            self.lib.lrd_shmx_close(self._shmx)
            self._shmx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
