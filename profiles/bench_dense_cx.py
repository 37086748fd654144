#!/usr/bin/env python3
"""Times the dense-objective kernel k_dense_cx (W = C X on the FP64 matrix cores) through the public
path (cal_obj on a dense-C cone): n = 4096, r = 34.  Run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
from lorads_amd import host  # noqa: E402

n, m = int(os.environ.get("N", 4096)), 800
rng = np.random.default_rng(7)
iu, ju = np.triu_indices(n)
g = rng.standard_normal(len(iu)) * 0.01
g[iu == ju] += 1.0
mat = [np.zeros(len(iu), np.int32)]
blk = [np.zeros(len(iu), np.int32)]
row, col, val = [iu.astype(np.int32)], [ju.astype(np.int32)], [-g]   # F0 = -C
b = np.zeros(m)
for k in range(m):
    i, j = rng.integers(0, n, 2)
    mat.append(np.array([k + 1], np.int32)); blk.append(np.zeros(1, np.int32))
    row.append(np.array([min(i, j)], np.int32)); col.append(np.array([max(i, j)], np.int32)); val.append(np.array([1.0]))
    b[k] = 0.1
s = host.Session.from_triplets(m, b, [n], np.concatenate(mat), np.concatenate(blk), np.concatenate(row), np.concatenate(col),
                               np.concatenate(val))
s.set_params(verbose=0, timesLogRank=float(os.environ.get("TLR", 4.0)))
s.prepare()
s.attach_hip()
info = s.block_info(0)
print("n", info["n"], "r", info["rank"], "dense_mode", info["dense_mode"])
for _ in range(3):
    s.be.cal_obj(host.PAIR_RR)
t0 = time.perf_counter()
K = 50
for _ in range(K):
    v = s.be.cal_obj(host.PAIR_RR)
dt = (time.perf_counter() - t0) / K
R = s.be.get_mat(host.MAT_R, 0)
C = np.zeros((n, n)); C[iu, ju] = g; C = C + C.T - np.diag(np.diag(C))
ref = float(np.sum(R * (C @ R)))
print("cal_obj %.1f us per call; value %.12e vs numpy %.12e" % (dt * 1e6, v, ref))
assert abs(v - ref) <= 1e-10 * abs(ref)
s.close()
