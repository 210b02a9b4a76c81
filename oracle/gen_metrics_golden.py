"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/graph_metrics_T40.npz by importing the REAL reference (/root/reference) in
this container through oracle/ref_shim.py:   PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python oracle/gen_metrics_golden.py

The host-side bookkeeping surface of the reference's I2cGraph that is not a message or a controller: the entropy / likelihood
metric lists _maximize fills (i2c.py:1004-1027), the observation covariances behind the temperature update (:913-919, 983-992),
the cost helpers (:1029-1066), the KL divergence (:1223-1229), get_prior_state_action_distribution (:1294-1298). Problem: the
covariance-control pendulum with a terminal cost of em_covctrl_qf_T40 (oracle/gen_golden.py), four EM iterations through
learn_msgs(). The file holds data only (inputs + the reference's outputs)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import numpy as np  # noqa: E402
from i2c.exp_types import CubatureQuadrature  # noqa: E402
from i2c.i2c import I2cGraph  # noqa: E402  (the reference)
from i2c.model import make_env_model  # noqa: E402

import gen_golden  # noqa: E402  (problem_inputs, save)


def main(T=40, n_iter=4):
    np.random.seed(3)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    mu_xt, sig_xt = np.array([0.5, 0.0]), np.array([[2e-3, 5e-4], [5e-4, 1e-2]])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), mu_xt, sig_xt, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = gen_golden.problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), mu_xt, sig_xt, (1, 0, 0),
                                    propagate=True, use_expert_controller=True, seed=3, n_iter=n_iter)
    g.propagate()
    for _ in range(n_iter):
        g.learn_msgs()
        g.calc_likelihood()
    for name in ("policy_entropy", "sig_eta_entropy", "sig_eta_pf_entropy", "x_prior_entropy", "x_prior_neg_entropy", "propagate_entropy",
                 "kl_terms", "costs_m", "costs_m_var", "costs_pf", "costs_pf_var", "cost_pf_min", "alphas", "alphas_desired", "alphas_pf",
                 "likelihoods", "likelihoods_xu", "likelihoods_z", "risk"):
        out[name] = np.asarray(getattr(g, name), dtype=float)
    out["cost_pf_entropy"] = np.asarray(g.cost_pf_entropy, dtype=float)
    out["propagate_cost_improved"] = np.asarray(float(g.propagate_cost_improved))
    z_covar, z_pf, z_term = g.get_z_covar(), g.get_z_propagated_covar(), g.get_z_terminal_covar()
    out["z_covar"], out["z_propagated_covar"], out["z_terminal_covar"] = z_covar, z_pf, z_term
    out["calculate_alpha"] = np.asarray(g.calculate_alpha(z_covar))
    out["calculate_alpha_term"] = np.asarray(g.calculate_alpha(z_covar, z_term))
    mu_p, sig_p = g.get_prior_state_action_distribution()
    out["prior_prev_mu"], out["prior_prev_sig"] = np.asarray(mu_p, float), np.asarray(sig_p, float)
    c = g.cells[5]
    out["cost_gaussian_in_mu"], out["cost_gaussian_in_sig"] = np.asarray(c.mu_xu0_m, float).reshape(-1), np.asarray(c.sig_xu0_m, float)
    out["cost_gaussian"] = np.asarray(g.compute_cost_gaussian(c.mu_xu0_m, c.sig_xu0_m), float)
    # (compute_cost(x, u), i2c.py:1029-1032, cannot run in the reference: it calls sys.observe with two arguments)
    m1, s1 = np.array([[0.1], [0.2]]), np.array([[0.5, 0.1], [0.1, 0.3]])
    out["kl_in"] = np.concatenate((m1.reshape(-1), s1.reshape(-1), mu_xt.reshape(-1), sig_xt.reshape(-1)))
    out["kl"] = np.asarray(g.mvn_kl_divergence(m1, s1, mu_xt.reshape(-1, 1), sig_xt), float)
    d, s, t = g.calc_sig_eta_entropy_max()
    out["sig_eta_entropy_max"] = np.concatenate(([d, t], np.asarray(s, float).reshape(-1)))
    d, s, t = g.calc_sig_eta_pf_entropy_max()
    out["sig_eta_pf_entropy_max"] = np.concatenate(([d, t], np.asarray(s, float).reshape(-1)))
    ok, s = g.calc_sig_eta_bound_check()
    out["sig_eta_bound_check"] = np.concatenate(([float(ok)], np.asarray(s, float).reshape(-1)))
    gen_golden.save("graph_metrics_T40", {k: np.asarray(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
