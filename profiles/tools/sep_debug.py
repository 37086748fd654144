"""two ranks on one card, verbose solver log per rank (debugging aid for the sharded forms): sep_debug.py <instance> <world> [key=value ...]"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, name, params):
    import torch
    import torch.distributed as dist
    import bench
    from tests import common
    faulthandler.dump_traceback_later(int(os.environ.get("SEP_DEBUG_DUMP_S", "60")), exit=True)
    sys.stdout = open("/tmp/sep_debug_rank%d.log" % rank, "w", buffering=1)
    os.dup2(sys.stdout.fileno(), 1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    s = common.hip_session(common.instance_path(name), world=world, rank=rank, **params)
    s.set_params(verbose=1)
    calls = [0]
    bench.install_allreduce(s, dist, torch, device, world, rank, "gloo")
    s.solve()
    r = s.results()
    print("RESULT", rank, r, flush=True)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    name, world = sys.argv[1], int(sys.argv[2])
    params = {}
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        params[k] = float(v) if "." in v or "e" in v else int(v)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, 29611, name, params)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    for p in procs:
        if p.is_alive():
            p.kill()
    for r in range(world):
        print("=== rank", r)
        lines = open("/tmp/sep_debug_rank%d.log" % r).read().splitlines()
        print("\n".join(lines[:8] + ["..."] + lines[-25:]))
