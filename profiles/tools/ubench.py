"""Kernel variants of the headline cone (rand20000: n = 20000, r = 40, 5000 constraints) launched back to back
through lorads_hip_ubench: microseconds per launch between two events.  Run it under
`rocprofv3 --kernel-trace --stats` to get every variant's own duration as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lorads_amd import host  # noqa: E402

NAMES = {0: "k_cw<8 lanes, two entries per trip>", 1: "k_cw<4 lanes, one trip>", 2: "k_spmm_ell (fixed-width slot list)",
         3: "k_spmm<CW> (CSR slot list)", 4: "k_spmm2<FRONT> (rhs + initial residual)", 5: "k_spmm2 (rhs only)",
         6: "k_cg_update 2048 wg", 7: "k_cg_update 1024 wg", 8: "k_cg_update 512 wg", 9: "k_cg_update 256 wg",
         10: "k_obj", 11: "k_sval (two images)"}


ENTRY = {100: "k_op_entry (whole operator, single-entry constraints)"}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else sorted(NAMES)
    workload = sys.argv[3] if len(sys.argv) > 3 else "rand20000"
    tlr = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
    NAMES.update(ENTRY)
    path = bench.build_instance(workload, "/tmp/lorads_bench_%s.dat-s" % workload)
    s = host.Session.open(path)
    s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
    s.prepare(1, 0)
    s.attach_hip()
    s.alm()
    s.alm_to_admm()
    s.be.init_constr(host.PAIR_UV)
    for w in which:
        ms = s.hip_ubench(w, reps)
        print("%2d  %-45s %8.2f us/launch" % (w, NAMES[w], 1e3 * ms / reps), flush=True)
    s.close()


if __name__ == "__main__":
    main()
