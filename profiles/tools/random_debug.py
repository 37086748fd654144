"""debugging aid: the ADMM part of tests/test_hip_random_instances.py for one seed under several A/B switches; prints the largest
difference of U / V per cone against the oracle:  random_debug.py <seed>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lorads_amd import host, instances  # noqa: E402
from tests import common  # noqa: E402
from tests.test_hip_random_instances import random_problem  # noqa: E402

seed = int(sys.argv[1])
prob = random_problem(7000 + seed)
path = "/tmp/lorads_random_dbg_%d.dat-s" % seed
instances.write_sdpa(prob, path)
tlr = [1.0, 2.0, 3.5][seed % 3]
print("seed", seed, "dims", prob["blocks"], "m", prob["m"])
SW = [{}, {"LORADS_NO_MERGE": "1"}, {"LORADS_NO_BATCH": "1"}, {"LORADS_FRONT_CW": "0"}, {"LORADS_LAZY_SCALARS": "0"}, {"LORADS_SPLIT_FRONT": "1"},
      {"LORADS_EXACT_REFRESH": "1"}, {"LORADS_OP_CW": "0"}, {"LORADS_OP_CW": "1"}, {"LORADS_NO_ELL": "1", "LORADS_NO_SLOT_ELL": "1"}, {"LORADS_FRONT_DIAG": "0"},
      {"LORADS_EVAL_DIAG": "0"}, {"LORADS_FOLD_AVG": "0"}, {"LORADS_FUSE_DIR": "0"}, {"LORADS_NO_OP_ENTRY": "1"}]
for sw in SW:
    for k, v in sw.items():
        os.environ[k] = v
    hs = common.hip_session(path, timesLogRank=tlr)
    for k in sw:
        os.environ.pop(k)
    os_ = common.oracle_session(path, timesLogRank=tlr)
    nb = hs.nblk
    fro2 = sum(v * v * (1.0 if i == j else 2.0) for (mat, blk, i, j, v) in prob["entries"] if mat > 0)
    rho2 = [0.3, 1.0, 4.0][(seed + 1) % 3]
    rng = np.random.default_rng(1)
    for k in range(nb):
        Rk = os_.be.get_mat(host.MAT_R, k)
        Rk = Rk / max(np.abs(Rk).max(), 1e-300) / (4.0 * np.sqrt(1.0 + fro2))
        for s in (hs, os_):
            s.be.set_mat(host.MAT_R, k, Rk)
    lam = rng.normal(size=hs.m)
    for s in (hs, os_):
        s.be.set_vec(host.VEC_LAMBDA, lam)
        s.be.alm_to_admm()
        s.be.init_constr(host.PAIR_UV)
    out = []
    for step in range(2):
        ia = hs.be.admm_update_var(rho2, 1e-10, 400)
        ib = os_.be.admm_update_var(rho2, 1e-10, 400)
        d = []
        for k in range(nb):
            for w in (host.MAT_U, host.MAT_V):
                x, y = hs.be.get_mat(w, k), os_.be.get_mat(w, k)
                d.append(float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300)))
        out.append((ia, ib, ["%.1e" % v for v in d]))
        for s in (hs, os_):
            s.be.update_dual_var(rho2)
        for k in range(nb):
            hs.be.set_mat(host.MAT_U, k, os_.be.get_mat(host.MAT_U, k))
            hs.be.set_mat(host.MAT_V, k, os_.be.get_mat(host.MAT_V, k))
        hs.be.set_vec(host.VEC_LAMBDA, os_.be.get_vec(host.VEC_LAMBDA))
        hs.be.set_vec(host.VEC_CONSTR_SUM, os_.be.get_vec(host.VEC_CONSTR_SUM))
    print(sw, [hs.hip_operator_kind(k) for k in range(nb)] if not sw else "")
    for o in out:
        print("   ", o)
    hs.close()
    os_.close()
