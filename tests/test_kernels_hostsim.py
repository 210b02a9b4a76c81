"""CPU-only check of the KERNEL MATH: csrc/ compiled as a host simulation (tests/hostsim.py)
and driven through the same C ABI + engine as on the GPU, against the golden vectors of the
real reference and against the batched CPU oracle. The GPU twins of these tests are in
tests/test_hip_parity.py."""
import pytest

import hostsim
import parity


@pytest.fixture(scope="module")
def lib():
    return hostsim.load()


GOLDEN = [
    ("em_pendulum_T200", 1e-9, 1e-8),
    ("em_pendulum_T40_quad_general", 1e-9, 1e-8),
    ("em_dcp_T60", 1e-7, 1e-6),
    ("em_cartpole_T100", 1e-7, 1e-6),
    ("em_linear_T60", 1e-9, 1e-8),
    ("em_covctrl_T100", 1e-8, 1e-7),
    ("em_pendulum_T50_propagate", 1e-9, 1e-8),
]


@pytest.mark.parametrize("name,tol_d,tol_s", GOLDEN)
def test_hostsim_vs_reference_golden(lib, name, tol_d, tol_s):
    parity.check_against_golden(name, lib, "cpu", tol_d, tol_s)


@pytest.mark.parametrize("name,B,iters", [("em_pendulum_T200", 8, 4), ("em_dcp_T60", 4, 3), ("em_covctrl_T100", 4, 4)])
def test_hostsim_batch_vs_oracle(lib, name, B, iters):
    parity.check_batch_against_oracle(name, lib, "cpu", B, iters, tol=1e-7)
