"""Whole solves with the NATIVE all-reduce hook on one rank (RCCL communicator of size 1): exercises every place the
sharded path calls the hook -- device buffers in stream order (ADMM evaluation, phase-1 dots and m-vectors) and host
buffers (rank agreement on decisions) -- against the same solves without a hook."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tests import common  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
dist.init_process_group("nccl", device_id=device)
bad = 0
for name, params in [("blk4x60", dict(reoptLevel=1, phase1Tol=1e-2)), ("rand120", dict(reoptLevel=1, phase1Tol=1e-2)),
                     ("maxcut100", dict(reoptLevel=2)), ("sdplp40", dict(reoptLevel=1, phase1Tol=1e-2))]:
    with common.hip_session(common.instance_path(name), **params) as s:
        s.solve()
        ref = s.results()
    for sep in (False, True):   # the m-vector form, then the scalars-only form of separable shards (one rank: trivially separable)
        s = common.hip_session(common.instance_path(name), world=1, rank=0, separable=sep, **params)
        assert s.separable == sep
        mode, _seen = bench.install_allreduce(s, dist, torch, device, 1, 0, "nccl")
        s.solve()
        got = s.results()
        nat = getattr(s, "_rccl_native", None)
        if nat:
            s.hip_sync()
            nat[0].lorads_rccl_comm_destroy(nat[1])
        s.close()
        # (the sharded phase 1 sums its dots over another partition of the vectors than the single-rank fused step: long runs
        # separate by rounding, as in tests/test_multirank_hip.py -- converged objectives to 2e-6)
        ok = all(abs(got[k] - ref[k]) <= 2e-6 * (1 + abs(ref[k])) for k in ("pObj", "dObj"))
        bad += not ok
        print(name, mode, "separable" if sep else "m-vector", "ok" if ok else "MISMATCH", got["pObj"], ref["pObj"], got["admm_iter"], ref["admm_iter"],
              got["alm_inner"], ref["alm_inner"])
dist.destroy_process_group()
sys.exit(1 if bad else 0)
