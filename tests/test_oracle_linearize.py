"""Pins the Linearize restatement (oracle/i2c_linearize_numpy.py) against vectors captured from the REAL reference
(oracle/gen_golden.py lin_* cases). CPU only. The linear cases need no Jacobian and pin the solver algebra exactly; the
nonlinear ones were captured with a complex-step stand-in for autograd.jacobian (not installed) -- see the oracle header."""
import numpy as np
import pytest

from golden_util import assert_close, load_case, oracle_from_case

FWD = ["mu_xu0_f", "sig_xu0_f", "mu_xu1_f", "sig_xu1_f", "mu_x3_f", "sig_x3_f", "J_dyn", "mu_z0_f", "sig_z0_f"]
BWD = ["mu_xu0_m", "sig_xu0_m", "K", "k", "sigK", "mu_z0_m", "sig_z0_m", "mu_x3_m", "sig_x3_m"]
PF = ["mu_xu0_pf", "sig_xu0_pf", "mu_x3_pf", "sig_x3_pf", "mu_z0_pf", "sig_z0_pf"]
LINS = {"A": "A", "B": "Bm", "a": "a", "E": "E", "F": "F"}


def run_and_check(name, tol_detail, tol_summary):
    g = load_case(name)
    o = oracle_from_case(g)
    detail = set(g.iters())
    n = len(g["costs_m"])
    for it in range(1, n + 1):
        o.em_iter += 1
        o.forward_sweep()
        if it in detail:
            for k in FWD:
                assert_close(getattr(o, k)[0], g.at(it, k), tol_detail, f"{name} it{it} {k}")
            for k, attr in LINS.items():
                assert_close(getattr(o, attr)[0], g.at(it, k), tol_detail, f"{name} it{it} {k}")
        o.backward_sweep()
        if it in detail:
            for k in BWD:
                assert_close(getattr(o, k)[0], g.at(it, k), tol_detail, f"{name} it{it} {k}")
            assert_close(o.mu_z3_m[0], g.at(it, "mu_z3_m"), tol_detail, f"{name} it{it} mu_z3_m")
            assert_close(o.sig_z3_m[0], g.at(it, "sig_z3_m"), tol_detail, f"{name} it{it} sig_z3_m")
        if o._propagate:
            o.propagate()
            if it in detail:
                for k in PF:
                    assert_close(getattr(o, k)[0], g.at(it, k), tol_detail, f"{name} it{it} {k}")
        o.maximize()
    assert_close(np.array(o.alphas)[:, 0], g["alphas"], tol_summary, name + " alphas")
    assert_close(np.array(o.alphas_desired)[:, 0], g["alphas_desired"], tol_summary, name + " alphas_desired")
    assert_close(np.array(o.costs_m)[:, 0], g["costs_m"], tol_summary, name + " costs_m")
    assert_close(np.array(o.costs_m_var)[:, 0], g["costs_m_var"], tol_summary, name + " costs_m_var")
    assert_close(np.array(o.costs_pf)[:, 0], g["costs_pf"], tol_summary, name + " costs_pf")
    if "kl_terms" in g:
        assert_close(np.array(o.kl_terms)[:, 0], g["kl_terms"], tol_summary * 10, name + " kl_terms")
    if "alphas_pf" in g:
        assert_close(np.array(o.alphas_pf)[:, 0], g["alphas_pf"], tol_summary, name + " alphas_pf")
    assert_close(o.K[0], g["final/K"], tol_summary, name + " final K")
    assert_close(o.k[0], g["final/k"], tol_summary, name + " final k")
    assert_close(o.sigK[0], g["final/sigK"], tol_summary, name + " final sigK")
    assert_close(o.mu_xu0_m[0], g["final/mu_xu0_m"], tol_summary, name + " final mu_xu0_m")
    assert_close(o.sig_xu0_m[0], g["final/sig_xu0_m"], tol_summary, name + " final sig_xu0_m")


@pytest.mark.parametrize("name,td,ts", [
    ("lin_linear_T60", 1e-8, 1e-7),
    ("lin_covctrl_T50", 1e-8, 1e-7),
    ("lin_covctrl_qf_T30", 1e-8, 1e-7),  # + a terminal cost: the back-calculated sig_xi_terminal (i2c.py:455-462)
    ("lin_pendulum_T100", 1e-8, 1e-6),
    ("lin_pendulum_T40_propagate", 1e-8, 1e-6),  # + closed-loop propagation, expert controller
    ("lin_cartpole_T100", 1e-8, 1e-6),
    ("lin_dcp_T80", 1e-7, 1e-6),
    ("lin_quad12_T20", 1e-8, 1e-6),  # 12-state quadrotor (d = 16): the reference's I2cGraph on the build-defined model
])
def test_linearize_em(name, td, ts):
    run_and_check(name, td, ts)


def test_lqr_compare_protocol_with_riccati_messages():
    """scripts/lqr_compare.py (config 0): one forward/backward pass with Linearize, then the Riccati messages; the
    Riccati-form controller is the reference's, and both are the finite-horizon LQR solution."""
    g = load_case("lin_lqr_compare")
    o = oracle_from_case(g)
    o.forward_backward()
    for k in FWD:
        assert_close(getattr(o, k)[0], g.at(1, k), 1e-7, "lqr " + k)
    for k in BWD:
        assert_close(getattr(o, k)[0], g.at(1, k), 1e-7, "lqr " + k)
    K_msg, k_msg, mu = o.K[0].copy(), o.k[0].copy(), o.mu_xu0_m[0].copy()
    o.riccati_sweep()
    assert_close(o.K[0], g["riccati/K"], 1e-7, "riccati K")
    assert_close(o.k[0], g["riccati/k"], 1e-7, "riccati k")
    assert_close(o.sigK[0], g["riccati/sigK"], 1e-7, "riccati sigK")
    assert_close(o.lambda_x0_b[0], g["riccati/lambda_x0_b"], 1e-7, "riccati lambda_x0_b")
    assert_close(o.nu_x0_b[0], g["riccati/nu_x0_b"], 1e-7, "riccati nu_x0_b")
    # LQR equivalence (the point of the script); the gains differ near the end, where the terminal cost acts
    h = K_msg.shape[0] // 2
    assert_close(K_msg[:h], g["lqr/K"][:h], 1e-6, "K vs LQR")
    assert_close(k_msg[:h], g["lqr/k"][:h], 1e-6, "k vs LQR")
    assert_close(mu[:, :2], g["lqr/x"][: mu.shape[0]], 1e-6, "x vs LQR")
    assert_close(mu[:, 2:], g["lqr/u"], 1e-5, "u vs LQR")
