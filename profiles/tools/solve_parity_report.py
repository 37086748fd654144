#!/usr/bin/env python3
"""Whole solves through the HIP path vs the reference's runs in tests/golden/solve.json: objectives, DIMACS numbers and
iteration counts side by side (diagnostic behind tests/test_hip_parity.py::test_whole_solve_vs_reference)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402


def params(flags):
    return {flags[i][2:]: float(flags[i + 1]) if "." in flags[i + 1] or "e" in flags[i + 1] else int(flags[i + 1])
            for i in range(0, len(flags), 2)}


for e in common.golden_solves():
    s = common.hip_session(common.instance_path(e["instance"]), **params(e["flags"]))
    try:
        r = s.solve()
    finally:
        s.close()
    dense = e["wsum_is_dense"] if isinstance(e["wsum_is_dense"], list) else [e["wsum_is_dense"]]
    rel = lambda a, b: abs(a - b) / (1 + abs(b))  # noqa: E731
    print("%-12s %-60s sparse=%d  pObj rel %.2e dObj rel %.2e | err1 %.2e (ref %.2e) gap %.2e (ref %.2e) | inner %d/%d admm %d/%d cg %d/%d"
          % (e["instance"], " ".join(e["flags"]), int(all(x == 0 for x in dense)), rel(r["pObj"], e["pObj"]), rel(r["dObj"], e["dObj"]),
             r["constrVio1"], e["err_constr_l1"], r["pdGap"], e["err_pdgap"], r["alm_inner"], e["alm_inner"], r["admm_iter"],
             e["admm_iter"], r["cg_iter"], e["admm_cg_iter"]), flush=True)
