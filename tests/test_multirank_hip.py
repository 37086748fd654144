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


@pytest.fixture(autouse=True)
def _several_processes_on_one_card(monkeypatch):
    """the ranks of these tests share the box's one GPU: their one-launch ADMM iterations (teams of resident workgroups,
    csrc/hip/persist.inc) take turns instead of waiting for each other's compute units (the workers inherit the variable)"""
    monkeypatch.setenv("LORADS_SHARED_GPU", "1")
    yield


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
        if os.environ.get("LORADS_TEST_SHMX"):   # the evaluation's scalars from host to host (lorads_hip_set_scalar_exchange)
            s.set_scalar_exchange_shm(os.environ["LORADS_TEST_SHMX"], world, rank)
        s.hip_profile(1, 1 << 30)   # (counts the solves resumed after a missed speculation)
        s.solve()
        r = s.results()
        r["exchanges"] = s.hip_scalar_exchange_count()
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
    ("mix4+misses", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, False),
    # the separable form with the evaluation's four scalars summed by the ranks' HOSTS through shared memory (no collective on the
    # stream in phase 2): whole solves with reopt rounds and rank growth, one cone per rank, and the miss protocol at work
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 2, "shm"),
    ("mix4", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, "shm"),
    ("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2), 4, "shm"),
    ("mix4+misses", dict(reoptLevel=1, phase1Tol=1e-2, phase2Tol=1e-7), 2, "shm")])
def test_two_ranks_on_device_match_single_process(built, monkeypatch, name, params, world, separable):
    """Sharded cones, one process per rank on the card.  separable=True: no constraint touches cones of two ranks, each rank works
    on the sub-problem over its own constraints and the ranks share scalars only (lorads_hip_set_separable: four doubles per ADMM
    iteration); False: the m-vector form."""
    from tests import common
    provoke = name.endswith("+misses")
    name = name.split("+")[0]
    shm = separable == "shm"
    separable = bool(separable)
    if shm:
        monkeypatch.setenv("LORADS_TEST_SHMX", "/lorads_test_%d_%s_%d" % (os.getpid(), name, world))
    else:
        monkeypatch.delenv("LORADS_TEST_SHMX", raising=False)
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
    if shm:     # phase 2's evaluations went from host to host (phase 1's collectives keep the hook)
        assert all(out[r]["exchanges"] >= a["admm_iter"] > 0 for r in range(world)), [out[r]["exchanges"] for r in range(world)]
    else:
        assert all(out[r]["exchanges"] == 0 for r in range(world))
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


def _worker_fullsize(rank, world, port, path, tlr, statefile, its, separable, q, shm=None):
    """one rank of test_fullsize_sharded_iterations...: the cones dealt to it, the single-process state loaded into them, `its`
    ADMM iterations with the cross-rank sums through the device-buffer all-reduce hook"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from lorads_amd import host
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    s = host.Session.open(path)
    s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
    s.prepare(world, rank, separable=separable)
    s.attach_hip()
    try:
        mode, seen = bench.install_allreduce(s, dist, torch, device, world, rank, "gloo")
        assert seen == world
        if shm:      # the evaluation's scalars from host to host through shared memory instead of through the hook
            s.set_scalar_exchange_shm(shm, world, rank)
        st = np.load(statefile)
        for j in range(s.nblk):                 # cones are dealt round robin: local cone j is cone rank + j * world of the file
            k = rank + j * world
            s.be.set_mat(host.MAT_U, j, st["U%d" % k])
            s.be.set_mat(host.MAT_V, j, st["V%d" % k])
        s.be.set_vec(host.VEC_LAMBDA, st["lam"][np.asarray(s.constraint_map)])
        rho = float(st["rho"])
        be = s.be
        be.init_constr(host.PAIR_UV)
        be.cal_obj(host.PAIR_UV)
        e0 = be.update_dimacs(host.PAIR_UV)
        e1, cg, p1, d1 = bench.admm_steps(be, host, rho, e0, its, s)
        s.hip_sync()
        Us = {"U%d" % (rank + j * world): be.get_mat(host.MAT_U, j) for j in range(s.nblk)}
        q.put((rank, dict(err0=e0, err1=e1, cg=int(cg), pObj=p1, dObj=d1, nblk_local=s.nblk, m_local=s.m, separable=bool(s.separable),
                          mode=mode, U=Us, exchanges=s.hip_scalar_exchange_count())))
    finally:
        s.close()
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,separable", [(2, True), (4, True), (2, False), (4, False), (2, "shm"), (4, "shm")])
def test_fullsize_sharded_iterations_match_one_process_and_the_reference(built, world, separable):
    """BASELINE cfg4 at full size -- 16 cones of n = 4000, 64000 constraints -- dealt over 2 and 4 ranks (processes on the one card,
    device-buffer all-reduce hook), in the scalars-only form of separable shards and in the m-vector form: 20 ADMM iterations from
    the SINGLE-PROCESS phase-1 state.  Block-separable constraints: the ranks' Jacobi sweep is the reference's Gauss-Seidel sweep
    (SURVEY.md 8e), so the sharded run must reproduce the single-process run iterate for iterate -- same CG iteration count, same
    objectives and factors up to the order of the cross-rank sums -- and the compiled reference's 20 iterations from the same state
    (oracle/_ref/ref_driver admmbench) to 1e-10."""
    import subprocess
    import numpy as np
    import bench
    from lorads_amd import host
    from tests import common
    from tests.test_hip_parity import _gen
    # "shm": the separable form with the evaluation's four scalars summed by the ranks' hosts through shared memory
    # (lorads_hip_set_scalar_exchange) -- no collective on the stream between two ADMM iterations
    shm = "/lorads_test_%d_%d" % (os.getpid(), world) if separable == "shm" else None
    separable = bool(separable)
    its, tlr = 20, 2.0
    path = _gen("blk16x4000")
    statefile = "/tmp/lorads_shard_state_%d.npz" % os.getpid()
    refstate = "/tmp/lorads_shard_state_%d.bin" % os.getpid()
    s = host.Session.open(path)
    s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
    s.prepare(1, 0)
    s.attach_hip()
    try:
        s.alm()
        s.alm_to_admm()
        res = s.results()
        rho = min(res["admm_rho"] if res["admm_rho"] > 0 else res["alm_rho"], 5000.0)
        be = s.be
        nb = s.nblk
        UV = [(be.get_mat(host.MAT_U, k), be.get_mat(host.MAT_V, k)) for k in range(nb)]
        lam = be.get_vec(host.VEC_LAMBDA)
        np.savez(statefile, rho=rho, lam=lam, **{"U%d" % k: UV[k][0] for k in range(nb)}, **{"V%d" % k: UV[k][1] for k in range(nb)})
        with open(refstate, "wb") as f:
            for U, V in UV:
                f.write(np.asfortranarray(U).tobytes(order="F"))
                f.write(np.asfortranarray(V).tobytes(order="F"))
            f.write(lam.tobytes())
        be.init_constr(host.PAIR_UV)
        be.cal_obj(host.PAIR_UV)
        e0 = be.update_dimacs(host.PAIR_UV)
        e1, cg1, p1, d1 = bench.admm_steps(be, host, rho, e0, its, s)
        U_one = [be.get_mat(host.MAT_U, k) for k in range(nb)]
        ranks = [s.block_shape(k)[1] for k in range(nb)]
    finally:
        s.close()
    ref = None
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv):
        env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", LORADS_REF_UV_RANKS=",".join(str(r) for r in ranks))
        r = subprocess.run([drv, path, "admmbench", "-", "--timesLogRank", repr(tlr), "--rho", repr(rho), "--uv", refstate, "--nADMM", str(its)],
                           env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("@@REF_ADMM_BENCH")]
        assert r.returncode == 0 and line, r.stderr[-800:]
        ref = dict(x.split("=") for x in line[0].split()[1:])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fullsize, args=(r, world, port, path, tlr, statefile, its, separable, q, shm)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = dict(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
        for f in (statefile, refstate):
            if os.path.exists(f):
                os.remove(f)
    a = out[0]
    if shm:     # every evaluation of the 20 iterations (and the two in front of them) went from host to host
        assert all(out[r]["exchanges"] >= its for r in range(world)), [out[r]["exchanges"] for r in range(world)]
    else:
        assert all(out[r]["exchanges"] == 0 for r in range(world))
    assert sum(out[r]["nblk_local"] for r in range(world)) == nb
    assert all(out[r]["separable"] == separable for r in range(world))
    for r in range(1, world):   # every rank holds the same summed scalars
        for k in ("pObj", "dObj", "err1", "err0"):
            assert a[k] == out[r][k], (r, k, a[k], out[r][k])
    cg = sum(out[r]["cg"] for r in range(world))
    print("blk16x4000 world", world, "separable" if separable else "m-vector", "cg", cg, int(cg1), "pObj", a["pObj"], p1, "dObj", a["dObj"], d1,
          "err1", a["err1"], e1, "reference", ref)
    assert cg == int(cg1), (cg, cg1)
    assert a["pObj"] == pytest.approx(p1, rel=1e-11) and a["dObj"] == pytest.approx(d1, rel=1e-11)
    assert a["err1"] == pytest.approx(e1, rel=1e-6, abs=1e-14)
    for r in range(world):
        for name, U in out[r]["U"].items():
            k = int(name[1:])
            assert np.allclose(U, U_one[k], rtol=0, atol=1e-9 * np.abs(U_one[k]).max()), (r, k)
    if ref:
        assert cg == int(ref["cg_iters"])
        assert a["pObj"] == pytest.approx(float(ref["pObj"]), rel=1e-10) and a["dObj"] == pytest.approx(float(ref["dObj"]), rel=1e-10)
