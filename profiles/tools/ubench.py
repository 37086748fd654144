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
         30: "k_front_cw (rhs + initial residual + slot contributions)", 31: "k_front_cw without the second visit of the slots",
         32: "k_wsum (constraint weights from the contributions)",
         45: "row-major operator application (k_cw + k_spmm_ell) + k_cg_update in between",
         10: "k_obj", 11: "k_sval (two images)", 13: "k_obj, up to 2048 workgroups", 14: "k_obj, up to 4096 workgroups",
         20: "gather probe: 2.56 M random 320-B rows of V (6.4 MB table)", 21: "gather probe: rows of x and V in turn (12.8 MB)",
         22: "gather probe: the same number of rows in ascending order",
         23: "gather probe, front-shaped: 16 random rows of V per row, 20000 rows", 24: "gather probe, front-shaped, x and V in turn",
         25: "front-shaped probe on the union neighbour list, 16 slots per row", 26: "  + the rows' real lists (5-35 slots)",
         27: "  + a coefficient gather per slot", 28: "  + own row read, full row written (= k_spmm2 without epilogue)"}
PROBE_BYTES = {20: 40000 * 64 * 320, 21: 40000 * 64 * 320, 22: 40000 * 64 * 320, 23: 20000 * 16 * 320, 24: 20000 * 16 * 320}


ENTRY = {100: "k_op_entry / k_op_entry_bip x 2 (whole operator, single-entry constraints; LORADS_ENTRY_BIP=0: the former)",
         120: "gather probe on this cone: 16 random rows of V per row", 121: "gather probe on this cone: rows of x and V in turn"}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else sorted(NAMES)
    workload = sys.argv[3] if len(sys.argv) > 3 else "rand20000"
    tlr = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
    NAMES.update(ENTRY)
    path = bench.build_instance(workload, "/tmp/lorads_bench_%s.dat-s" % workload)
    s = host.Session.open(path)
    s.set_params(verbose=0, timesLogRank=tlr, phase1Tol=1e-2, reoptLevel=0)
    if workload == "matcomp50000":
        s.set_params(dyrankLevel=0)   # (r = 60 as BASELINE cfg5 names it; with rank growth phase 1 ends at r = 90)
    s.prepare(1, 0)
    s.attach_hip(libpath=host.DEV_LIB)   # (lorads_hip_ubench lives in the development build only)
    s.alm()
    s.alm_to_admm()
    s.be.init_constr(host.PAIR_UV)
    for w in which:
        ms = s.hip_ubench(w, reps)
        pb = dict(PROBE_BYTES)
        if w in (120, 121):
            n_, r_ = s.block_shape(0)
            pb[w] = n_ * 16 * r_ * 8
        extra = "  = %.2f TB/s of rows" % (pb[w] / (1e-3 * ms / reps) / 1e12) if w in pb else ""
        print("%2d  %-45s %8.2f us/launch%s" % (w, NAMES[w], 1e3 * ms / reps, extra), flush=True)
    s.close()


if __name__ == "__main__":
    main()
