"""bench.py's plumbing that needs no GPU: the replication of a block for the weak-scaling workload, the source stamp that decides
whether a committed profile may be replayed into the JSON line, and the choice of the committed profile files."""
import json
import os
import sys

import numpy as np

from tests import common

sys.path.insert(0, common.ROOT)
import bench  # noqa: E402


def test_replicate_blocks_is_block_separable(tmp_path):
    src = common.instance_path("maxcut100")
    out = str(tmp_path / "x3.dat-s")
    bench.replicate_blocks(src, 3, out)
    with open(src) as f:
        a = f.read().split("\n")
    with open(out) as f:
        b = f.read().split("\n")
    m, n = int(a[0]), int(a[2].split()[0])
    assert int(b[0]) == 3 * m and int(b[1]) == 3 and b[2].split() == [str(n)] * 3
    assert len(b[3].split()) == 3 * m
    ents = [ln.split() for ln in b[4:] if ln.strip()]
    # constraint i of copy k is constraint k m + i and lives in block k + 1 only: block-separable
    for e in ents:
        mat, blk = int(e[0]), int(e[1])
        if mat > 0:
            assert (mat - 1) // m == blk - 1
    assert len(ents) == 3 * len([ln for ln in a[4:] if ln.strip()])


def test_profile_stamp_gates_the_replay(tmp_path, monkeypatch):
    h = bench.hip_source_hash()
    assert len(h) == 64 and h == bench.hip_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "hip_source_hash", lambda: h)
    (prof / "r07_pmc_rand20000.json").write_text(json.dumps({"cg_operator_application": {"traffic_bytes": 1.0}}))
    assert bench.profile_stamp_ok("profiles/r07_pmc_rand20000.json") == (False, None)      # no stamp: not replayed
    (prof / "r07_stamp.json").write_text(json.dumps({"hip_source_sha256": "0" * 64, "git_head": "abc"}))
    assert bench.profile_stamp_ok("profiles/r07_pmc_rand20000.json")[0] is False           # other sources: stale
    (prof / "r07_stamp.json").write_text(json.dumps({"hip_source_sha256": h, "git_head": "abc"}))
    assert bench.profile_stamp_ok("profiles/r07_pmc_rand20000.json") == (True, "abc")
    assert bench.traffic_from_profiles("rand20000") == ({"cg_operator_application": {"traffic_bytes": 1.0}},
                                                        os.path.join("profiles", "r07_pmc_rand20000.json"))


def test_kernel_profile_is_the_default_runs(tmp_path, monkeypatch):
    """the committed rocprofv3 summary that is replayed is the one of the default run (the kernels as the timed iterations launch them)"""
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    hdr = '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
    (prof / "r07_rand20000_kernel_stats.csv").write_text(hdr + '"void k_wsum(int)",110,1,5000.0,1,1,1,0\n"void k_spmm_ell<8>(int)",110,1,9000.0,1,1,1,0\n"void k_front_cw<3, 16>(int)",110,1,27000.0,1,1,1,0\n')
    (prof / "r07_general_form_rand20000_kernel_stats.csv").write_text(hdr + '"void k_cw<4>(int)",169,1,12000.0,1,1,1,0\n"void k_spmm_ell<8>(int)",110,1,10000.0,1,1,1,0\n')
    ms, src = bench.rocprof_from_profiles("rand20000", "k_wsum+k_spmm_ell")
    assert src.endswith("r07_rand20000_kernel_stats.csv") and np.isclose(ms, 0.014)
    ms, src = bench.rocprof_from_profiles("rand20000", "k_front_cw")
    assert np.isclose(ms, 0.027)
