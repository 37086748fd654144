"""bench.py's plumbing that needs no GPU: the replication of a block for the weak-scaling workload, the source stamp that decides
whether a committed profile may be replayed into the JSON line, and the choice of the committed profile files."""
import json
import os
import sys

import numpy as np
import pytest

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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_line_keeps_the_contract_end_to_end():
    """`python bench.py` as the driver runs it, on a small workload: ONE JSON line on stdout with the contract's keys, the two objects
    the tier asks for (`roofline`, `cpu_baseline`), and round 4's `phase1` / `launches_per_step`"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--workload", "maxcut800",
                        "--times-log-rank", "2.0", "--cpu-budget", "3", "--no-extra", "--roofline-samples", "20"],
                       capture_output=True, text=True, timeout=550, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "phase1", "launches_per_step"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert d["launches_per_step"] == 1.0      # (a Max-Cut-type context: the whole iteration is one launch)
    p1 = d["phase1"]
    assert p1["inner_iters"] > 0 and p1["us_per_inner_iter"] > 0 and p1["launches_per_inner_iter"] < 12
    assert p1["again_in_a_warm_process"]["inner_iters"] == p1["inner_iters"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("args,scaling", [(["--workload", "rand120"], "weak"), (["--scaling", "strong", "--workload", "blk4x60"], "strong")])
def test_bench_line_with_two_ranks_on_one_card(args, scaling):
    """`python bench.py --gpus 2` starting its own ranks (two processes on the one card, gloo as the hook's transport): the line names
    two ranks SEEN through the library's hook, carries the sharded-parity block, and the process exits 0 only because both hold"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LORADS_DIST_BACKEND="gloo", LORADS_FORCE_DEVICE="0", MASTER_PORT="29611")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu", "--no-extra",
                        "--roofline-samples", "0"] + args, capture_output=True, text=True, timeout=800, cwd=root, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == scaling, (d["n_gpus"], d["ranks_seen"], d["scaling"])
    ps = d["parity_sharded"]
    assert ps["ok"] is True and ps["cg_iters_sharded"] == ps["cg_iters_single_rank"] and ps["pObj_rel_diff"] <= 1e-10, ps
    assert d["value"] > 0 and d["scalar_exchange"] is not None
