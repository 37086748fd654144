"""CPU-side checks: the C-ABI library loads and exports every symbol include/*.h declares (no compute
without a GPU), the product refuses to run without the HIP backend (no CPU fallback), the host's own
reader / pre-solve / scalar code."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

from lorads_amd import host, instances
from tests import common

ROOT = common.ROOT


def test_abi_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "lorads_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lorads_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 28
    lib = C.CDLL(os.path.join(host.LIB_DIR, "liblorads_hip.so"))
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # the measurement / diagnostic entries live in their own header and are exported too; the drop-in header has none of them
    dev = open(os.path.join(ROOT, "include", "lorads_hip_dev.h")).read()
    dev_syms = sorted(set(re.findall(r"\b(lorads_hip_[a-z0-9_]+)\s*\(", dev)))
    assert {"lorads_hip_profile", "lorads_hip_profile_samples", "lorads_hip_time_operator", "lorads_hip_operator_kind"} <= set(dev_syms)
    # (lorads_hip_ubench -- single kernel variants, probe kernels -- exists in the development build only: the product library must NOT have it)
    assert not [s for s in dev_syms if not hasattr(lib, s) and s != "lorads_hip_ubench"]
    assert not hasattr(lib, "lorads_hip_ubench"), "the product library carries the development build's ubench"
    assert not (set(dev_syms) & set(declared)), "measurement entries leaked into the product header"


def test_rccl_hook_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "lorads_rccl.h")).read()
    declared = sorted(set(re.findall(r"\b(lorads_rccl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) == 6, declared
    lib = C.CDLL(os.path.join(host.LIB_DIR, "liblorads_rccl.so"))
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # no RCCL named and none on the default path that is not loadable -> an error code and a message, not a crash
    lib.lorads_rccl_last_error.restype = C.c_char_p
    if lib.lorads_rccl_open(b"/nonexistent/librccl.so") != 0:
        assert b"dlopen" in lib.lorads_rccl_last_error()


def test_host_library_exports_table_adaptor(built):
    lib = host.host_lib()
    for s in ("lrd_hip_backend_create", "lrd_read_sdpa", "lrd_alm_optimize", "lrd_admm_optimize", "lrd_solve", "lrd_reopt"):
        assert hasattr(lib, s)
    lib.lrd_backend_sizeof.restype = C.c_size_t
    assert lib.lrd_backend_sizeof() == C.sizeof(host.BackendStruct)  # ctypes mirror == C struct


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_product_fails_loudly_without_gpu(built):
    s = host.Session.open(common.instance_path("maxcut100"))
    try:
        s.set_params(verbose=0)
        s.prepare()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            s.attach_hip()
    finally:
        s.close()


def _write(tmp_path, text):
    p = tmp_path / "x.dat-s"
    p.write_text(text)
    return str(p)


def test_reader_conventions(tmp_path, oracle_lib):
    # comments, braces/commas in the dimension and rhs lines, upper/lower triangle, tiny entries, F0 negated
    path = _write(tmp_path, '"a comment\n* another\n3 = m\n2\n{3, 2}\n{1.0, 2.5, -3}\n'
                            "0 1 1 2 4.0\n0 1 3 3 -1.0\n0 2 1 1 2.0\n1 1 1 1 1.0\n1 1 2 1 0.5\n2 2 2 1 7.0\n2 2 1 1 1e-13\n3 1 3 3 2.0\n")
    s = host.Session.open(path, lib=oracle_lib)
    try:
        s.set_params(verbose=0)
        s.prepare()
        assert (s.m, s.nblk) == (3, 2)
        b0, b1 = s.block_info(0), s.block_info(1)
        assert (b0["n"], b0["nrow"], b0["na"], b0["nc"]) == (3, 2, 3, 2)   # constraints 1 and 3 touch block 1
        assert (b1["n"], b1["nrow"], b1["na"], b1["nc"]) == (2, 1, 1, 1)   # the 1e-13 entry is dropped
        assert b0["dense_mode"] == 1 and b1["dense_mode"] == 1            # n < 20 -> dense scratch
        assert b0["cone_sparse"] == 0 and b1["cone_sparse"] == 0           # 2/3 and 1/3 > 0.3 m
    finally:
        s.close()


def test_reader_lp_block(tmp_path, oracle_lib):
    """One diagonal (LP) block is accepted when it is the last block (io/lorads_file_io.c:120-124): it becomes a
    cone image with is_lp semantics -- n = number of columns, rank 1, diagonal entries keyed by their row index.
    Anywhere else, or of dimension 0, it is refused."""
    path = _write(tmp_path, "2\n2\n2 -3\n1.0 2.0\n0 2 2 2 -0.5\n1 1 1 1 1.0\n1 2 1 1 1.0\n2 2 3 3 2.0\n2 2 1 1 0.5\n")
    s = host.Session.open(path, lib=oracle_lib)
    try:
        s.set_params(verbose=0)
        s.prepare()
        assert (s.m, s.nblk) == (2, 2)
        lp = s.block_info(1)
        assert (lp["n"], lp["rank"], lp["nrow"], lp["na"], lp["nc"], lp["dense_mode"]) == (3, 1, 2, 3, 1, 0)
    finally:
        s.close()
    for bad in ("1\n2\n-3 2\n1.0\n1 2 1 1 1.0\n", "1\n2\n2 0\n1.0\n1 1 1 1 1.0\n"):
        with pytest.raises(RuntimeError):
            host.Session.open(_write(tmp_path, bad), lib=oracle_lib)


def test_rank_rule_and_branches(tmp_path, oracle_lib):
    # data/lorads_solver.c:290-319: r = min(ceil(t log n), floor(sqrt(2 #A))+1, n)
    for name, t, expect in (("maxcut100", 2.0, min(math.ceil(2 * math.log(100)), int(math.sqrt(200)) + 1)),
                            ("maxcut800", 2.0, 14), ("maxcut800", 4.0, 27)):
        s = host.Session.open(common.instance_path(name), lib=oracle_lib)
        try:
            s.set_params(verbose=0, timesLogRank=t)
            s.prepare()
            assert s.block_info(0)["rank"] == expect
        finally:
            s.close()


def test_separable_deal_is_cut_down_to_the_ranks_own_constraints(oracle_lib):
    """lrd_problem_localize: blk4x60 (every constraint lives in one cone) dealt over 2 and 4 ranks leaves each rank the
    sub-problem over its own constraints -- a partition of the file's constraints, b and the norms of the whole problem kept;
    coupled3x70 (constraints that touch several cones) is left alone."""
    full = host.Session.open(common.instance_path("blk4x60"), lib=oracle_lib)
    full.set_params(verbose=0)
    full.prepare()
    m_all, rows_all = full.m, [full.block_info(k)["nrow"] for k in range(full.nblk)]
    full.close()
    for world in (2, 4):
        seen = []
        for rank in range(world):
            s = host.Session.open(common.instance_path("blk4x60"), lib=oracle_lib)
            try:
                s.set_params(verbose=0)
                s.prepare(world, rank, separable=True)
                assert s.separable and s.m_global == m_all and s.nblk == 4 // world
                assert s.m == sum(rows_all[k] for k in range(4) if k % world == rank)
                cm = s.constraint_map
                assert len(cm) == s.m and np.all(np.diff(cm) > 0)      # ascending file order
                seen.append(cm.copy())
            finally:
                s.close()
        allc = np.sort(np.concatenate(seen))
        assert np.array_equal(allc, np.arange(m_all))                   # a partition
    s = host.Session.open(common.instance_path("coupled3x70"), lib=oracle_lib)
    try:
        s.set_params(verbose=0)
        s.prepare(2, 0, separable=True)
        assert not s.separable and s.m == s.m_global
    finally:
        s.close()
    s = host.Session.open(common.instance_path("blk4x60"), lib=oracle_lib)
    try:
        s.set_params(verbose=0)
        s.prepare(2, 1)                                                 # not asked for: the general sharded form
        assert not s.separable and s.m == m_all
    finally:
        s.close()


def test_cubic_and_linesearch_scalars(oracle_lib):
    lib = oracle_lib
    lib.lrd_cubic_roots.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double)]
    r = (C.c_double * 3)()
    # (x-1)(x-2)(x-3) = x^3 - 6x^2 + 11x - 6 : three real roots
    n = lib.lrd_cubic_roots(1.0, -6.0, 11.0, -6.0, r)
    assert n == 3 and np.allclose(sorted(r), [1, 2, 3], atol=1e-9)
    # x^3 + x = 0 has the single real root 0 (delta > 0 branch)
    n = lib.lrd_cubic_roots(1.0, 0.0, 1.0, 0.0, r)
    assert n == 1 and abs(r[0]) < 1e-12
    # quartic t^4 - t^2: minimiser on (0,1] is 1/sqrt(2)
    tau, nroot = common.linesearch_tau([1.0, 0.0, -1.0, 0.0])
    assert nroot == 3 and abs(tau - 1 / math.sqrt(2)) < 1e-12
    # convex increasing quartic: stay at 0
    tau, _ = common.linesearch_tau([1.0, 0.0, 1.0, 1.0])
    assert tau == 0.0


def test_generators_are_deterministic(tmp_path):
    a, b = tmp_path / "a", tmp_path / "b"
    for p in (a, b):
        instances.write_sdpa(instances.NAMED["rand120"](), str(p))
    assert a.read_bytes() == b.read_bytes()
    assert a.read_bytes() == open(common.instance_path("rand120"), "rb").read()  # the committed fixture


def test_separate_and_fused_host_paths_agree(oracle_lib):
    """the oracle table has no fused slot, so the host loop takes the 4-call path; the session switch exists"""
    s = common.oracle_session(common.instance_path("maxcut100"), reoptLevel=0)
    try:
        assert not s.be.has_admm_step
        s.use_fused_step(0)
        r = s.solve()
        assert int(r["admm_iter"]) == 5
    finally:
        s.close()
