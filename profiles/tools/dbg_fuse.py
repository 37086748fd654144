import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import common
from lorads_amd import host
name = sys.argv[1] if len(sys.argv) > 1 else "blk4x60"
g = common.golden_trace(name)
res = []
for fuse in ("1", "0"):
    os.environ["LORADS_FUSE_DIR"] = fuse
    s = common.hip_session(common.instance_path(name))
    os.environ.pop("LORADS_FUSE_DIR")
    rank_warm = [int(x) for x in g["rank_warm"]]
    if rank_warm != [s.block_shape(k)[1] for k in range(s.nblk)]:
        s.be.resize_rank(rank_warm)
    for k in range(s.nblk):
        n, r = s.block_shape(k)
        s.be.set_mat(host.MAT_R, k, g["R_warm_0_%d" % k].reshape(r, n).T)
    s.be.set_vec(host.VEC_LAMBDA, g["lambda_warm"])
    s.be.alm_to_admm(); s.be.init_constr(host.PAIR_UV); s.be.cal_obj(host.PAIR_UV); s.be.update_dimacs(host.PAIR_UV)
    rho = float(g["admm_rho"][0])
    log = []
    for it, tol in enumerate([1e-8, 1e-8, 1e-12, 1e-6, 1e-10, 1e-9]):
        c, p, d, e = s.be.admm_step(rho, tol, 800)
        log.append((c, p, [s.be.get_mat(host.MAT_U, k).copy() for k in range(s.nblk)]))
        s.be.update_dual_var(rho)
    res.append(log)
    s.close()
for it, (a, b) in enumerate(zip(*res)):
    print(it, a[0], b[0], a[1], b[1], [float(np.abs(x - y).max()) for x, y in zip(a[2], b[2])])
