import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU restatement (oracle/liblorads_oracle.so) -- the checker, never the product."""
    from tests.common import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def built():
    """Make sure the product libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    return True


# Tests that compare two forms of the LAUNCH-BY-LAUNCH ADMM iteration with each other (a switch against the default, bit for bit or
# launch count for launch count): they keep their meaning only if both sides really run launch by launch, so the one-launch form of
# Max-Cut-type cones (csrc/hip/persist.inc) is off for them.  That form has its own A/B tests (tests/test_persist.py), and every test
# that compares the product with the reference or the oracle runs the default, i.e. the one-launch form wherever it applies.
LAUNCH_BY_LAUNCH_TESTS = {
    "test_fused_step_equals_separate_calls", "test_fused_alm_step_equals_separate_calls", "test_lockstep_sweep_equals_cone_by_cone",
    "test_recurrence_and_fused_front_equal_two_pass_form", "test_one_kernel_front_equals_coefficient_pass_plus_fused_front",
    "test_scalar_steps_on_carrier_kernels_are_bitwise_the_separate_launches",
    "test_lockstep_convergence_test_on_the_last_workgroup_is_bitwise_its_own_launch",
    "test_lockstep_carriers_at_iteration_limits_and_later_restarts", "test_carried_scalar_steps_on_a_grid_larger_than_the_device",
    "test_fused_front_of_maxcut_cones_equals_the_separate_passes", "test_graph_replay_is_bitwise_the_launch_by_launch_iteration",
    "test_fused_paths_equal_the_step_by_step_forms",
}
# ... and the tests that compare phase 1's fused calls with the slot-by-slot calls bit for bit: the one-launch L-BFGS direction of
# the fused step (csrc/hip/lbfgs_team.inc, its own A/B tests: tests/test_lbfgs_team.py) sums its dots in another order
STAGE_BY_STAGE_TESTS = {"test_fused_alm_step_equals_separate_calls", "test_alm_front_and_step_function_level"}


@pytest.fixture(autouse=True)
def _launch_by_launch_form(request, monkeypatch):
    name = getattr(request.node, "originalname", request.node.name)
    if name in LAUNCH_BY_LAUNCH_TESTS:
        monkeypatch.setenv("LORADS_PERSIST", "0")
    if name in STAGE_BY_STAGE_TESTS:
        monkeypatch.setenv("LORADS_LBFGS_TEAM", "0")
    yield
