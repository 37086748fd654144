"""csrc/host/shmx.c: the host-to-host sum of a few doubles over the ranks of one node (the scalar exchange of separable shards,
lorads_hip_set_scalar_exchange).  CPU: several processes, many calls, every rank must see the same bits; a rank that never arrives
ends the others' wait with an error, not a hang."""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    sys.path.insert(0, ROOT)
    from lorads_amd import host
    lib = host.host_lib()
    lib.lrd_shmx_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.lrd_shmx_allreduce.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.lrd_shmx_close.argtypes = [C.c_void_p]
    return lib


def _worker(rank, world, name, calls, q, skip_last):
    lib = _lib()
    h = C.c_void_p()
    if lib.lrd_shmx_open(name.encode(), world, rank, C.byref(h)) != 0:
        q.put((rank, "open failed"))
        return
    rng = np.random.default_rng(1000 + rank)
    digest = 0.0
    out = []
    rc = 0
    for c in range(calls):
        n = 1 + (c % 7)
        mine = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8)
        if skip_last and rank == world - 1 and c == calls - 1:
            break                                   # this rank leaves before the last call: the others must time out
        v = (C.c_double * n)(*mine)
        rc = lib.lrd_shmx_allreduce(h, v, n)
        if rc:
            break
        out.append(np.array(v[:n]))
        if c % 97 == 0:
            time.sleep(0.0005 * rank)               # ranks drift apart now and then
    q.put((rank, rc, [o.tobytes() for o in out]))
    if skip_last and rank == world - 1:
        time.sleep(3.0)                             # (keep the segment alive while the others run into their time limit)
    lib.lrd_shmx_close(h)


@pytest.mark.parametrize("world", [2, 4])
def test_every_rank_sees_the_same_sums(world):
    calls = 3000
    name = "/lorads_test_%d_%d" % (os.getpid(), world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, name, calls, q, False)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict((r, (rc, out)) for r, rc, out in [q.get(timeout=600) for _ in range(world)])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(res[r][0] == 0 and len(res[r][1]) == calls for r in range(world))
    for r in range(1, world):
        assert res[r][1] == res[0][1]               # bit for bit
    # ... and they are the sums in rank order
    rngs = [np.random.default_rng(1000 + r) for r in range(world)]
    for c in range(calls):
        n = 1 + (c % 7)
        acc = np.zeros(n)
        for r in range(world):
            acc = acc + rngs[r].standard_normal(n) * 10.0 ** rngs[r].integers(-8, 8)
        assert np.frombuffer(res[0][1][c]).tobytes() == acc.tobytes(), c
    assert not os.path.exists("/dev/shm" + name)    # rank 0 has removed the segment


def test_a_rank_that_never_arrives_ends_the_wait(monkeypatch):
    monkeypatch.setenv("LORADS_HANDOVER_TIMEOUT_S", "1")
    world, calls = 3, 50
    name = "/lorads_test_%d_late" % os.getpid()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, name, calls, q, True)) for r in range(world)]
    t0 = time.time()
    for p in ps:
        p.start()
    res = dict((r, (rc, out)) for r, rc, out in [q.get(timeout=600) for _ in range(world)])
    for p in ps:
        p.join(timeout=120)
    assert time.time() - t0 < 40
    assert res[world - 1][0] == 0 and len(res[world - 1][1]) == calls - 1
    for r in range(world - 1):
        assert res[r][0] == 1 and len(res[r][1]) == calls - 1     # the last call failed, after the time limit


def test_bad_arguments():
    lib = _lib()
    h = C.c_void_p()
    assert lib.lrd_shmx_open(b"no-slash", 2, 0, C.byref(h)) != 0
    assert lib.lrd_shmx_open(b"/lorads_test_bad", 2, 2, C.byref(h)) != 0
    assert lib.lrd_shmx_open(("/lorads_test_one_%d" % os.getpid()).encode(), 1, 0, C.byref(h)) == 0
    v = (C.c_double * 3)(1.5, -2.0, 1e-300)
    assert lib.lrd_shmx_allreduce(h, v, 3) == 0 and list(v) == [1.5, -2.0, 1e-300]
    assert lib.lrd_shmx_allreduce(h, v, 17) != 0
    lib.lrd_shmx_close(h)


def _leftover_maker(name, world, q):
    """a 'rank 0' of an earlier run that dies without closing: its segment stays behind under the name"""
    lib = _lib()
    h = C.c_void_p()
    q.put(lib.lrd_shmx_open(name.encode(), world, 0, C.byref(h)))
    q.close()
    q.join_thread()      # (the answer must be out before the process goes -- without closing, let alone unlinking, anything)
    os._exit(0)


def _late_rank0(name, world, delay, q):
    time.sleep(delay)
    lib = _lib()
    h = C.c_void_p()
    rc = lib.lrd_shmx_open(name.encode(), world, 0, C.byref(h))
    v = (C.c_double * 1)(1.0)
    rc2 = lib.lrd_shmx_allreduce(h, v, 1) if rc == 0 else 1
    q.put(("r0", rc, rc2, v[0]))
    lib.lrd_shmx_close(h)


def test_a_leftover_segment_of_a_dead_run_is_not_attached_to(monkeypatch, request):
    """advisor r3: a run that died leaves a valid-looking segment of the same name (uid + rendezvous port).  A rank that opens the
    name before the new rank 0 has replaced it must not settle on the leftover (its maker is gone): it waits for the fresh one."""
    monkeypatch.setenv("LORADS_HANDOVER_TIMEOUT_S", "5")
    name = "/lorads_test_%d_leftover" % os.getpid()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    dead = ctx.Process(target=_leftover_maker, args=(name, 2, q))
    dead.start()
    request.addfinalizer(lambda: os.path.exists("/dev/shm" + name) and os.unlink("/dev/shm" + name))
    assert q.get(timeout=300) == 0
    dead.join(timeout=120)
    assert os.path.exists("/dev/shm" + name)          # the leftover is there, magic word and all
    r0 = ctx.Process(target=_late_rank0, args=(name, 2, 1.0, q))
    r0.start()                                         # the new rank 0 arrives a second AFTER rank 1 has started to open
    lib = _lib()
    h = C.c_void_p()
    assert lib.lrd_shmx_open(name.encode(), 2, 1, C.byref(h)) == 0
    v = (C.c_double * 1)(2.0)
    assert lib.lrd_shmx_allreduce(h, v, 1) == 0 and v[0] == 3.0    # the two ranks met on the SAME (fresh) segment
    tag, rc, rc2, got = q.get(timeout=300)
    assert (rc, rc2, got) == (0, 0, 3.0)
    lib.lrd_shmx_close(h)
    r0.join(timeout=120)
