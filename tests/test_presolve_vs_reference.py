"""The pre-solve's decisions against the reference's own (tests/golden/presolve.json, written by oracle/make_golden.py from
`ref_driver presolve`): AConeProcData / AConePresolveData / LORADSDetermineRank / LUserDataChooseCone
(data/lorads_sdp_conic.c:868-1076, data/lorads_sdp_data.c:811-828, data/lorads_solver.c:290-319, io/lorads_user_data.c:58).
CPU: the host half (csrc/host/problem.c).  GPU: the image lorads_hip_create builds from it (csrc/hip/build.inc)."""
import json
import os

import numpy as np
import pytest

from lorads_amd import host, instances
from tests import common

GOLD = json.load(open(os.path.join(common.GOLD, "presolve.json")))
CASES = sorted(GOLD)


def _entries(name):
    """per cone: objective positions, per-constraint positions (lower-triangular, duplicates merged) from the generator's entries"""
    prob = instances.NAMED[name]()
    cones = [dict(n=abs(d), lp=d < 0, C=set(), A={}) for d in prob["blocks"]]
    for mat, blk, i, j, v in prob["entries"]:
        if abs(v) < 1e-14:       # (the reader drops tiny entries, io/lorads_file_io.c:250)
            continue
        pos = (max(i, j) - 1, min(i, j) - 1)
        if mat == 0:
            cones[blk - 1]["C"].add(pos)
        else:
            cones[blk - 1]["A"].setdefault(mat - 1, set()).add(pos)
    return prob, cones


@pytest.mark.parametrize("case", CASES)
def test_host_presolve_matches_the_reference(case):
    name, tlr = case.split("@")
    ref = GOLD[case]
    s = host.Session.open(common.instance_path(name))
    s.set_params(verbose=0, timesLogRank=float(tlr))
    s.prepare(1, 0)
    try:
        assert s.m == ref["m"]
        sdp = [k for k in range(s.nblk)]
        assert len(ref["cones"]) + (1 if ref["nlp"] else 0) == s.nblk
        for k, rc in enumerate(ref["cones"]):
            info = s.block_info(sdp[k])
            assert info["n"] == rc["n"] and info["rank"] == rc["rank"], (k, info, rc)
            assert info["cone_sparse"] == rc["cone_sparse"], (k, info, rc)
            assert info["dense_mode"] == rc["wsum_dense"], (k, info, rc)
            assert info["np"] == rc["wsum_nnz"] == rc["objsum_nnz"], (k, info, rc)       # union pattern of C and the A_i (or n(n+1)/2)
            # constraints with an entry in the cone: what a sparse cone holds; a dense cone holds all m, the others as zero matrices
            zero_constr = rc["n_zero"] - (1 if rc["obj_type"] == 0 else 0)
            assert info["nrow"] == (rc["rows_held"] if rc["cone_sparse"] else rc["rows_held"] - zero_constr), (k, info, rc)
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c.endswith("@2.0")])
def test_device_image_matches_the_reference(built, case):
    name, tlr = case.split("@")
    ref = GOLD[case]
    prob, cones = _entries(name)
    s = common.hip_session(common.instance_path(name), timesLogRank=float(tlr))
    try:
        for k, rc in enumerate(ref["cones"]):
            im = s.hip_block_image(k)
            cn = cones[k]
            assert im["n"] == rc["n"] and im["rank"] == rc["rank"]
            # coefficient types by the reference's rule (dense iff nnz > 0.1 n(n+1)/2); the device keeps a dense constraint full only
            # on cones of 32 rows and more (build.inc), and the objective wherever the rule says so
            assert im["dense_c"] == (1 if rc["obj_type"] == 2 else 0), (im, rc)
            ref_dense_a = rc["n_dense"] - (1 if rc["obj_type"] == 2 else 0)
            assert im["dense_a"] == (ref_dense_a if rc["n"] >= 32 else 0), (im, rc)
            thr = 0.1 * rc["n"] * (rc["n"] + 1) / 2
            dense_ids = {i for i, p in cn["A"].items() if len(p) > thr} if rc["n"] >= 32 else set()
            sparse_a = set().union(*[p for i, p in cn["A"].items() if i not in dense_ids]) if cn["A"] else set()
            assert im["pattern_a"] == len(sparse_a), (im["pattern_a"], len(sparse_a))
            union = sparse_a | (set() if im["dense_c"] else cn["C"])
            assert im["pattern_union"] == len(union)
            if not rc["wsum_dense"]:      # the reference's sparse scratch matrix IS that union pattern
                assert im["pattern_union"] == rc["wsum_nnz"]
            # operator kind from the constraints' shapes
            held = [p for i, p in sorted(cn["A"].items()) if i not in dense_ids]
            all_diag1 = bool(held) and not dense_ids and all(len(p) == 1 and next(iter(p))[0] == next(iter(p))[1] for p in held)
            all_single = bool(held) and not dense_ids and all(len(p) == 1 for p in held)
            assert im["diag_only"] == int(all_diag1), (im, name, k)
            assert im["entry_only"] == int(all_single and not all_diag1), (im, name, k)
            if im["use_cw"]:
                assert not im["diag_only"] and not im["entry_only"] and not im["dense_a"] and im["nrow"] >= 256
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c.endswith("@2.0")])
def test_device_sorts_build_the_host_patterns_small(built, case, monkeypatch):
    """The pattern work on the device (csrc/hip/presolve.inc: radix sorts + scans for the unique positions, the adjacency, the
    transpose of the constraint CSR) against the host construction it replaces, array by array, on every golden instance -- LP
    blocks, dense coefficients, empty cones, one-row cones included.  LORADS_PRESOLVE_CHECK=1 makes lorads_hip_create build both
    and refuse any difference; LORADS_DEV_PRESOLVE_MIN=0 sends even these small patterns through the device code."""
    monkeypatch.setenv("LORADS_PRESOLVE_CHECK", "1")
    monkeypatch.setenv("LORADS_DEV_PRESOLVE_MIN", "0")
    name, tlr = case.split("@")
    s = common.hip_session(common.instance_path(name), timesLogRank=float(tlr))
    try:
        st = s.hip_presolve_stats()
        assert st["device"] >= 2 and st["checked"] == st["device"], st      # (A-pattern and union pattern of every cone)
        # ... and nothing the construction launched was refused (an empty pattern must not become an empty grid): the next result
        # hand-over looks at the runtime's last error
        s.be.init_constr(host.PAIR_UV)
        assert np.isfinite(s.be.update_dimacs(host.PAIR_UV))
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["rand20000", "maxcut20000", "blk16x4000", "matcomp50000"])
def test_device_sorts_build_the_host_patterns_fullsize(built, name, monkeypatch):
    """The same comparison at BASELINE's sizes, with the default threshold (these patterns are the ones the device builds in a
    normal run), and the device image that comes out equals the one of a host-built context."""
    from tests.test_hip_parity import _gen
    path = _gen(name)
    images = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("LORADS_PRESOLVE_CHECK", "1" if mode == "device" else "0")
        monkeypatch.setenv("LORADS_DEV_PRESOLVE", "1" if mode == "device" else "0")
        if name == "blk16x4000":       # (16 cones of ~10^4 stored entries each: below the default threshold)
            monkeypatch.setenv("LORADS_DEV_PRESOLVE_MIN", "4096")
        s = common.hip_session(path)
        try:
            st = s.hip_presolve_stats()
            if mode == "device":
                assert st["device"] >= 1 and st["checked"] == st["device"], st      # (Max-Cut's A-pattern is its diagonal: host)
            else:
                assert st["device"] == 0
            images[mode] = [s.hip_block_image(k) for k in range(s.nblk)]
        finally:
            s.close()
    assert images["device"] == images["host"]
