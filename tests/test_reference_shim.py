"""The literal drop-in boundary: the REFERENCE'S OWN outer loops (LORADS_ALMOptimize, LORADSADMMOptimize, reopt, unmodified,
compiled from /root/reference where it lies) drive liblorads_hip.so through the operator table that
integration/lorads_func_hip.c installs by symbol interposition (oracle/Makefile ref_hip -> oracle/_ref_hip/, built in the build
container and shipped to the GPU box like oracle/_ref/).  The results are compared with the reference's CPU runs of the same
command (tests/golden/solve.json)."""
import os
import subprocess

import pytest
import torch

from tests import common

SHIM_DIR = os.path.join(common.ROOT, "oracle", "_ref_hip")
DRV = os.path.join(SHIM_DIR, "ref_driver_hip")
LIB = os.path.join(SHIM_DIR, "liblorads_func_hip.so")

TAKEN_OVER = ["LORADSInitFuncSet", "LORADSInitConstrValAll", "LORADSInitConstrValSum", "LORADSUpdateDualVar", "LORADSCalDualObj",
              "ALMLineSearch", "LORADS_ALMtoADMM", "objScale_dualvar", "AUG_RANK"]


def _need_shim():
    if not (os.path.exists(DRV) and os.path.exists(LIB)):
        pytest.skip("oracle/_ref_hip not built (compiled against /root/reference in the build container)")


def test_shim_defines_the_reference_symbols_it_takes_over(built):
    _need_shim()
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = [s for s in TAKEN_OVER if s not in defined]
    assert not missing, missing
    # the driver binds them from the shim: it is the first library the loader searches
    needed = subprocess.run(["readelf", "-d", DRV], capture_output=True, text=True, check=True).stdout
    libs = [ln.split("[")[1].split("]")[0] for ln in needed.splitlines() if "NEEDED" in ln]
    assert libs.index("liblorads_func_hip.so") < libs.index("liblorads_ref.so"), libs


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_shim_refuses_to_run_without_the_device(built):
    """no CPU path behind the table: the reference's loop must not spin on slots that compute nothing"""
    _need_shim()
    r = subprocess.run([DRV, common.instance_path("maxcut100"), "solve", "-", "--reoptLevel", "0"], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, MKL_NUM_THREADS="1"))
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    assert "MI355X backend is required" in r.stderr


def _golden(instance, flags):
    for e in common.golden_solves():
        if e["instance"] == instance and e["flags"] == flags:
            return e
    raise KeyError((instance, flags))


@pytest.mark.gpu
@pytest.mark.parametrize("instance,flags", [
    ("maxcut100", ["--reoptLevel", "0"]),                                   # sparse pattern branch, phase 1 + 5 ADMM iterations
    ("maxcut100", ["--reoptLevel", "1", "--phase1Tol", "1e-2"]),
    ("blk4x60", ["--reoptLevel", "1", "--phase1Tol", "1e-2"]),              # four sparse cones (merged cone / lockstep sweep)
    ("sdplp40", ["--reoptLevel", "0"]),                                     # SDP cone + LP block
    ("rand120", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),   # reopt round: objScale_dualvar, ALM_reopt, ADMM_reopt
    ("mix4", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),      # rank growth (AUG_RANK) inside the reference's loop
    ("densea40", ["--reoptLevel", "1", "--phase1Tol", "1e-2", "--phase2Tol", "1e-7"]),  # sdp_coeff_dense constraints flattened; MFMA path
])
def test_reference_loops_drive_the_hip_library(built, instance, flags):
    _need_shim()
    e = _golden(instance, flags)
    dump = "/tmp/lorads_shim_%d.bin" % os.getpid()
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", LORADS_REF_ALLOW_LP="1")
    r = subprocess.run([DRV, common.instance_path(instance), "solve", dump] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-600:])
    g = common.read_dump(dump)
    os.remove(dump)
    pobj, dobj = float(g["pObj"][0]), float(g["dObj"][0])
    ref_gap = abs(e["pObj"] - e["dObj"]) / (1 + abs(e["pObj"]) + abs(e["dObj"]))
    tol = max(2e-6, 5 * ref_gap)
    print(instance, flags, "pObj", pobj, e["pObj"], "dObj", dobj, e["dObj"], "inner", g["alm_inner"][0], e["alm_inner"], "admm",
          g["admm_iter"][0], e["admm_iter"], "cg", g["admm_cg_iter"][0], e["admm_cg_iter"], "final rank", g["final_rank"], e["final_rank"])
    assert abs(pobj - e["pObj"]) <= tol * (1 + abs(e["pObj"])), (pobj, e["pObj"])
    assert abs(dobj - e["dObj"]) <= tol * (1 + abs(e["dObj"])), (dobj, e["dObj"])
    assert float(g["err_constr_l1"][0]) <= max(10 * e["err_constr_l1"], 1e-5)
    fr = e["final_rank"] if isinstance(e["final_rank"], list) else [e["final_rank"]]
    assert [float(x) for x in g["final_rank"]] == [float(x) for x in fr]
    dense = e["wsum_is_dense"] if isinstance(e["wsum_is_dense"], list) else [e["wsum_is_dense"]]
    if all(x == 0 for x in dense):
        # sparse-mode instance: up to the first reopt round (phase 1, then the first ADMM pass) the reference's loops take the same
        # decisions on the device's numbers as on its own -- inner iterations of phase 1, ADMM iterations and CG iterations of the
        # first pass (rand120 with its reopt round is held to this too)
        assert int(g["alm_end_inner"][0]) == int(e["alm_end_inner"]), (g["alm_end_inner"][0], e["alm_end_inner"])
        assert int(g["admm_first_iter"][0]) == int(e["admm_first_iter"]), (g["admm_first_iter"][0], e["admm_first_iter"])
        assert int(g["admm_first_cg"][0]) == int(e["admm_first_cg"]), (g["admm_first_cg"][0], e["admm_first_cg"])
    if all(x == 0 for x in dense) and g["reopt_rounds"][0] == 0 and e["reopt_rounds"] == 0:
        # sparse-mode instance, no reopt round: the reference's loop takes the same decisions on the device's numbers
        assert int(g["alm_inner"][0]) == int(e["alm_inner"])
        assert int(g["admm_iter"][0]) == int(e["admm_iter"])
