import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests import common
import __graft_entry__
__graft_entry__.build()
NAMES = ["maxcut100", "theta30", "rand120", "blk4x60", "coupled3x70", "densec40", "densea40", "matcomp60", "mix4", "sdplp40", "sdpslack30", "coupledlp"]
for name in NAMES:
    g = common.golden_trace(name)
    s = common.hip_session(common.instance_path(name))
    try:
        log = common.replay_trace(s, g, rtol=1e-9, resync=True)
        dense = bool(np.any(np.asarray(g.get("wsum_is_dense", [0.0])) > 0))
        def worst(pred):
            v = [e for n_, e in log if pred(n_)]
            return max(v) if v else 0.0
        ph1 = worst(lambda n_: not (n_.startswith(("U_", "V_", "csum_uv", "admm_", "csum_rr", "lambda_", "cg_")) ) or n_ == "lambda_alm")
        uv = worst(lambda n_: n_.startswith(("U_", "V_")))
        vecs = worst(lambda n_: n_.startswith(("csum_uv", "csum_rr", "lambda_")) and n_ != "lambda_alm")
        obj = worst(lambda n_: n_.startswith(("admm_pobj", "admm_dobj")))
        err = worst(lambda n_: n_.startswith("admm_err1"))
        print("%-12s dense=%d phase1 %.2e  UV %.2e  vecs %.2e  obj %.2e  err1 %.2e" % (name, dense, ph1, uv, vecs, obj, err), flush=True)
    finally:
        s.close()
