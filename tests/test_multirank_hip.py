"""N > 1 on the device path: two processes (gloo rendezvous, both on cuda:0 -- the box has one GPU) each hold half
of the cones of a block-diagonal problem in their own HIP context; the cross-rank sums go through the all-reduce
hook on DEVICE buffers exactly as bench.py --gpus N does with RCCL.  With two cones per rank the lockstep sweep
runs inside each rank.  Compared with the single-process device run and the reference's golden solve."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, params, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    from tests import common
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    s = common.hip_session(common.instance_path(name), world=world, rank=rank, **params)
    try:
        mode, seen = bench.install_allreduce(s, dist, torch, device, world, rank, "gloo")
        assert seen == world, ("ranks seen by the library's own hook", seen, world)
        s.hip_profile(1, 1 << 30)   # (counts the solves resumed after a missed speculation)
        s.solve()
        r = s.results()
        r["speculation_misses"] = int(s.hip_profile_read()["speculation_misses"])
        r["nblk_local"] = s.nblk
        r["mode"] = mode
        r["ranks_seen"] = seen
        r["separable"] = s.separable
        r["m_local"] = s.m
        q.put((rank, r))
    finally:
        s.close()
        dist.barrier()
        dist.destroy_process_group()


# world = 4 on four cones: ONE cone per rank, the block-per-GPU layout of bench.py --gpus N -- the evaluation then sends the
# objective partials through the all-reduce and keeps the staged sums zeroed (no k_zero / k_stage_tail around the
# collective); the parent's own session makes it five processes on the card (the box allows six)
@pytest.mark.timeout(300)
@pytest.mark.parametrize("name,params,world,separable", [
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 2, True),
    ("mix4", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, True),
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 4, True),
    # the general sharded form (constrValSum, q1, q2 summed over the ranks as m-vectors): forced on a separable deal, and the only
    # form there is when constraints touch cones of two ranks (coupled3x70)
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 2, False),
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 4, False),
    ("coupled3x70", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, False),
    # speculation window 1 (enqueue what the previous sweep needed): a miss every other iteration -- the "somebody missed, everybody
    # evaluates again" protocol of both forms at work (the default window's runs above already miss at different iterations per rank)
    ("mix4+misses", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, True),
    ("mix4+misses", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, False)])
def test_two_ranks_on_device_match_single_process(built, monkeypatch, name, params, world, separable):
    """Sharded cones, one process per rank on the card.  separable=True: no constraint touches cones of two ranks, each rank works
    on the sub-problem over its own constraints and the ranks share scalars only (lorads_hip_set_separable: four doubles per ADMM
    iteration); False: the m-vector form."""
    from tests import common
    provoke = name.endswith("+misses")
    name = name.split("+")[0]
    with common.hip_session(common.instance_path(name), **params) as s:
        s.solve()
        ref = s.results()
        m_all, nb_all = s.m, s.nblk
    monkeypatch.setenv("LORADS_SEPARABLE", "1" if separable else "0")
    if provoke:
        monkeypatch.setenv("LORADS_SPEC_WINDOW", "1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, params, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = dict(q.get(timeout=150) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:  # a rank stuck in a collective must not outlive the test
            if p.is_alive():
                p.kill()
    a = out[0]
    assert sum(out[r]["nblk_local"] for r in range(world)) == nb_all
    assert all(out[r]["separable"] == separable for r in range(world))
    if separable:
        assert sum(out[r]["m_local"] for r in range(world)) == m_all       # the ranks' constraints partition the file's
    else:
        assert all(out[r]["m_local"] == m_all for r in range(world))
    assert a["admm_iter"] > 0, "phase 2 (the sharded evaluation) did not run"
    misses = [out[r]["speculation_misses"] for r in range(world)]
    print(name, "speculation misses per rank", misses)
    if provoke:
        assert sum(misses) > 50, misses   # (the ranks really missed, far more often than with the default window)
    for r in range(1, world):
        for k in ("pObj", "dObj", "constrVio1", "pdGap", "admm_iter", "alm_inner"):
            assert a[k] == out[r][k], (r, k, a[k], out[r][k])
    print(name, world, "separable" if separable else "m-vector form", "pObj", a["pObj"], ref["pObj"], "dObj", a["dObj"], ref["dObj"],
          "admm", a["admm_iter"], ref["admm_iter"], "alm inner", a["alm_inner"], ref["alm_inner"])
    # block-separable constraints: Jacobi across ranks == Gauss-Seidel; only summation orders differ.  Coupled
    # constraints: same fixed point, another trajectory.
    gold = [g for g in common.golden_solves() if g["instance"] == name and "1" in g["flags"][1:2]]
    assert a["pObj"] == pytest.approx(ref["pObj"], rel=2e-6)
    assert a["dObj"] == pytest.approx(ref["dObj"], rel=2e-6)
    assert a["constrVio1"] <= 1e-5 and a["pdGap"] <= 1e-5
    assert gold, "no reference solve of this instance in tests/golden/solve.json"
    e = gold[-1]  # the reference's own run of the same file (possibly at a tighter phase2Tol): converged objective
    gap = abs(e["pObj"] - e["dObj"]) / (1 + abs(e["pObj"]) + abs(e["dObj"]))
    assert abs(a["pObj"] - e["pObj"]) <= max(2e-5, 10 * gap) * (1 + abs(e["pObj"]))


@pytest.mark.timeout(300)
def test_native_rccl_hook_whole_solves_on_one_rank(built):
    """liblorads_rccl.so (ncclAllReduce in stream order on the library's stream, RCCL bound with dlopen) as the all-reduce
    hook of whole solves on an RCCL communicator of size 1 (the box has one GPU): every call site of the hook -- device
    buffers of the ADMM evaluation and of phase 1, host buffers of the rank agreement -- against the same solves
    without a hook; each in the m-vector form and in the scalars-only form of separable shards.  Runs as a child process (it initialises torch.distributed with the nccl backend)."""
    import subprocess
    tool = os.path.join(ROOT, "profiles", "tools", "native_hook_solve.py")
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=280, env=env)
    lines = [ln for ln in r.stdout.splitlines() if " rccl-native " in ln or " MISMATCH " in ln or " stream-ordered " in ln or " sync " in ln]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert len(lines) == 8 and all(" rccl-native " in ln and " ok " in ln for ln in lines), lines
    assert sum(" separable " in ln for ln in lines) == 4
