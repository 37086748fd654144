#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
prescribes) of bench.py into per-kernel HBM-side traffic per launch for the ADMM part of the run.

Corrections applied (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.  Infinity-Cache hits are
counted by these fabric-side counters, so "traffic" is L2-miss traffic, an upper bound of HBM bytes.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <workload> [alm]
"""
import collections
import csv
import json
import re
import statistics
import sys


# kernels only phase 1 launches: the ADMM part of a bench run starts behind the last of them (k_his_two*: the launch-by-launch inner
# iteration; k_lbfgs_team / k_alm_close: round 4's fused one)
PHASE1_MARKS = ("k_his_two", "k_lbfgs_team", "k_alm_close")


def short(n):
    m = re.search(r"(k_\w+|__amd_\w+)", n)
    return m.group(1) if m else n[:40]


ALM = len(sys.argv) > 5 and sys.argv[5] == "alm"   # the phase-1 part of the run (up to its last kernel) instead of the ADMM part


def per_kernel(path, counter):
    rows = list(csv.DictReader(open(path)))
    last = max(i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in PHASE1_MARKS))
    agg = collections.defaultdict(list)
    for r in (rows[:last + 1] if ALM else rows[last + 1:]):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[4], "part": "phase 1 (BM / ALM inner iterations)" if ALM else "ADMM iterations", "unit": "bytes per launch", "corrections": "KiB->bytes; FETCH_SIZE x2 (gfx950); WRITE_SIZE x1",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0.0])
    w = write.get(k, [0.0])
    out["kernels"][k] = {"launches": len(f), "read_bytes_mean": 2 * 1024 * sum(f) / len(f), "read_bytes_median": 2 * 1024 * statistics.median(f),
                         "write_bytes_mean": 1024 * sum(w) / len(w)}
ks = out["kernels"]
if "k_admm_diag" in ks:
    # every cone of Max-Cut type: the whole ADMM iteration is this one launch (csrc/hip/persist.inc)
    out["admm_iteration_one_launch"] = {"kernels": ["k_admm_diag"], "traffic_bytes": ks["k_admm_diag"]["read_bytes_median"] + ks["k_admm_diag"]["write_bytes_mean"]}
if "k_op_diag" in ks:
    op = ks["k_op_diag"]["read_bytes_median"] + ks["k_op_diag"]["write_bytes_mean"]
    out["cg_operator_application"] = {"kernels": ["k_op_diag"], "traffic_bytes": op}
elif "k_op_entry_bip" in ks:
    # two launches per application (the two colours of the entry graph): per-launch means over both sides, times two
    op = 2 * (ks["k_op_entry_bip"]["read_bytes_mean"] + ks["k_op_entry_bip"]["write_bytes_mean"])
    out["cg_operator_application"] = {"kernels": ["k_op_entry_bip", "k_op_entry_bip"], "traffic_bytes": op}
elif "k_op_entry" in ks:
    op = ks["k_op_entry"]["read_bytes_median"] + ks["k_op_entry"]["write_bytes_mean"]
    out["cg_operator_application"] = {"kernels": ["k_op_entry"], "traffic_bytes": op}
elif "k_front_cw" in ks and "k_wsum" in ks and "k_spmm_ell" in ks:
    # constraint-wise operator AS THE DEFAULT RUN APPLIES IT (iteration 0 behind the one-kernel front): k_wsum + k_spmm_ell
    op = sum(ks[k]["read_bytes_median"] + ks[k]["write_bytes_mean"] for k in ("k_wsum", "k_spmm_ell"))
    out["cg_operator_application"] = {"kernels": ["k_wsum", "k_spmm_ell"], "traffic_bytes": op}
    out["solve_front"] = {"kernels": ["k_front_cw"], "traffic_bytes": ks["k_front_cw"]["read_bytes_median"] + ks["k_front_cw"]["write_bytes_mean"]}
elif "k_cw" in ks and "k_spmm_ell" in ks:
    # constraint-wise operator, general form: k_cw (constraint values from the factors) + k_spmm_ell (fixed-width slot list)
    op = sum(ks[k]["read_bytes_median"] + ks[k]["write_bytes_mean"] for k in ("k_cw", "k_spmm_ell"))
    out["cg_operator_application"] = {"kernels": ["k_cw", "k_spmm_ell"], "traffic_bytes": op}
    if "k_spmm2" in ks:  # front of a solve (right-hand side + initial residual in one pass)
        out["solve_front"] = {"kernels": ["k_spmm2"], "traffic_bytes": ks["k_spmm2"]["read_bytes_median"] + ks["k_spmm2"]["write_bytes_mean"]}
    if "k_front_cw" in ks:  # the one-kernel front (+ k_wsum: iteration 0's constraint weights)
        out["solve_front"] = {"kernels": ["k_front_cw"], "traffic_bytes": ks["k_front_cw"]["read_bytes_median"] + ks["k_front_cw"]["write_bytes_mean"]}
        if "k_wsum" in ks:
            out["iteration0_weights"] = {"kernels": ["k_wsum"], "traffic_bytes": ks["k_wsum"]["read_bytes_median"] + ks["k_wsum"]["write_bytes_mean"]}
elif "k_cw" in ks and "k_spmm" in ks:
    # constraint-wise operator: k_cw (constraint values from the factors) + k_spmm<CW> CSR form (the more frequent k_spmm
    # population -> median)
    op = sum(ks[k]["read_bytes_median"] + ks[k]["write_bytes_mean"] for k in ("k_cw", "k_spmm"))
    out["cg_operator_application"] = {"kernels": ["k_cw", "k_spmm"], "traffic_bytes": op}
elif "k_spmm" in ks:
    # the CG operator uses the A-pattern adjacency (the smaller of the two k_spmm populations -> median)
    op = sum(ks[k]["read_bytes_median"] + ks[k]["write_bytes_mean"] for k in ("k_pairdots", "k_sgram", "k_spmm") if k in ks)
    out["cg_operator_application"] = {"kernels": [k for k in ("k_pairdots", "k_sgram", "k_spmm") if k in ks], "traffic_bytes": op}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out.get("cg_operator_application")))
