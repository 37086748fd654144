#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference through
oracle/_ref, see oracle/Makefile `make ref`).  What is committed is DATA only:

  tests/golden/<name>.dat-s          the generated input (oracle/gen_instances.py, fixed seeds)
  tests/golden/<name>.trace.npz      inputs/outputs of every lorads_func-table call of a scripted
                                     sequence (oracle/ref_driver.c, mode `trace`)
  tests/golden/solve.json            final objectives / DIMACS errors / iteration counts / log lines of
                                     whole solves (mode `solve`) for several flag sets
  tests/golden/presolve.json         what the reference's pre-solve decided for every cone of every instance (mode
                                     `presolve`: cone type, constraints held, scratch-matrix type and pattern size,
                                     coefficient matrices by type, rank); `python oracle/make_golden.py presolve`
                                     writes this file alone

usage: python oracle/make_golden.py            (from the repo root)
"""
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import gen_instances  # noqa: E402

TRACE = {
    # name: (nALM, nADMM, extra flags)
    "maxcut100": (8, 3, ["--phase1Tol", "1e-2"]),
    "theta30": (8, 3, ["--phase1Tol", "1e-2"]),
    "rand120": (8, 3, ["--phase1Tol", "1e-2"]),
    "blk4x60": (8, 3, ["--phase1Tol", "1e-2"]),
    "coupled3x70": (8, 3, ["--phase1Tol", "1e-2"]),
    "densec40": (8, 3, ["--phase1Tol", "1e-2"]),
    "matcomp60": (8, 3, ["--phase1Tol", "1e-2"]),
    "densea40": (8, 3, ["--phase1Tol", "1e-2"]),
    "mix4": (8, 3, ["--phase1Tol", "1e-2"]),
    "sdplp40": (8, 3, ["--phase1Tol", "1e-2"]),
    "sdpslack30": (8, 3, ["--phase1Tol", "1e-2"]),
    "coupledlp": (8, 3, ["--phase1Tol", "1e-2"]),
}
SOLVE = [
    ("maxcut100", ["--reoptLevel", "0"]),
    ("maxcut100", ["--reoptLevel", "1", "--phase1Tol", "1e-2"]),
    ("blk4x60", ["--reoptLevel", "0"]),
    ("blk4x60", ["--reoptLevel", "1", "--phase1Tol", "1e-2"]),
    ("theta30", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("theta50", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("rand120", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("coupled3x70", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("densec40", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("matcomp60", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("densea40", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("mix4", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("sdplp40", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("sdplp40", ["--reoptLevel", "0"]),
    ("sdpslack30", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("coupledlp", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),
    ("maxcut800", ["--reoptLevel", "0"]),
    ("maxcut800", ["--reoptLevel", "0", "--phase1Tol", "1e-2"]),
    # north-star wording: converged objectives to 1e-6 relative -- runs that converge well below that
    ("maxcut800", ["--reoptLevel", "1", "--phase2Tol", "1e-8"]),                       # cfg2 look-alike
    # (theta50, the cfg1 look-alike, does not get that far in the reference: 20000 ADMM iterations and a gap of 5e-5 at
    #  --phase2Tol 1e-8 -- its golden stays the 1e-7 run above, compared at the level the reference reached)
    ("maxcut800", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-8"]),
    ("rand120", ["--reoptLevel", "1", "--phase2Tol", "1e-8"]),
    ("blk4x60", ["--reoptLevel", "1", "--phase2Tol", "1e-8"]),
]


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from("<i", data, pos)
        pos += 4
        name = data[pos:pos + ln].decode()
        pos += ln
        (n,) = struct.unpack_from("<q", data, pos)
        pos += 8
        out[name] = np.frombuffer(data, dtype="<f8", count=n, offset=pos).copy()
        pos += 8 * n
    return out


def run_ref(args):
    env = dict(os.environ, MKL_NUM_THREADS="1", LORADS_REF_ALLOW_LP="1")
    r = subprocess.run([os.path.join(HERE, "_ref", "ref_driver")] + args, env=env, capture_output=True, text=True,
                       timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("ref_driver failed: %s\n%s" % (args, r.stderr[-2000:]))
    return r.stdout


def presolve_goldens(names):
    out = {}
    for n in names:
        for tlr in ("2.0", "4.0"):
            txt = run_ref([os.path.join(GOLD, n + ".dat-s"), "presolve", "-", "--timesLogRank", tlr])
            cones = []
            for ln in txt.splitlines():
                if ln.startswith("@@REF_PRESOLVE "):
                    cones.append({k: int(v) for k, v in (x.split("=") for x in ln.split()[1:])})
                elif ln.startswith("@@REF_PRESOLVE_END"):
                    tail = {k: int(v) for k, v in (x.split("=") for x in ln.split()[1:])}
            out["%s@%s" % (n, tlr)] = dict(tail, cones=cones)
        print("presolve", n, len(cones), "cone(s)")
    with open(os.path.join(GOLD, "presolve.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    os.makedirs(GOLD, exist_ok=True)
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    names = sorted(set(TRACE) | {n for n, _ in SOLVE})
    if sys.argv[1:] == ["presolve"]:   # (the instances are there already: this file alone)
        presolve_goldens(names)
        return
    for n in names:
        path = os.path.join(GOLD, n + ".dat-s")
        gen_instances.write_sdpa(gen_instances.NAMED[n](), path)
    presolve_goldens(names)
    for n, (nalm, nadmm, extra) in TRACE.items():
        dump = "/tmp/_golden_%s.bin" % n
        run_ref([os.path.join(GOLD, n + ".dat-s"), "trace", dump, "--nALM", str(nalm), "--nADMM", str(nadmm)] + extra)
        rec = read_dump(dump)
        rec["_nALM"] = np.array([nalm], dtype=np.float64)
        rec["_nADMM"] = np.array([nadmm], dtype=np.float64)
        np.savez_compressed(os.path.join(GOLD, n + ".trace.npz"), **rec)
        os.remove(dump)
        print("trace", n, len(rec), "records")
    solves = []
    for n, flags in SOLVE:
        dump = "/tmp/_golden_solve.bin"
        out = run_ref([os.path.join(GOLD, n + ".dat-s"), "solve", dump] + flags)
        rec = read_dump(dump)
        os.remove(dump)
        logs = [re.sub(r"\s+Time:.*$", "", ln) for ln in out.splitlines() if ln.startswith(("ALM OuterIter", "ADMM Iter"))]
        entry = dict(instance=n, flags=flags, log=logs)
        for k, v in rec.items():
            if v.size == 1 and not k.endswith("seconds"):
                entry[k] = float(v[0])
            elif k in ("rank", "final_rank", "cone_is_sparse", "wsum_is_dense"):
                entry[k] = [float(x) for x in v]
        solves.append(entry)
        print("solve", n, flags, entry["pObj"], entry["dObj"])
    with open(os.path.join(GOLD, "solve.json"), "w") as f:
        json.dump(solves, f, indent=1)


if __name__ == "__main__":
    main()
