"""Config 0 (scripts/lqr_compare.py): linear-Gaussian i2c against the closed-form finite-horizon
LQR solution -- an independent known answer (textbook Riccati recursion, written here from scratch).

The reference makes this comparison with its Linearize path; its cubature path cannot, because
LinearDef ships sig_x0 = sig_eta = 1e-20 (env_def.py:160-161) and `sum w y y^T - m m^T` cancels
catastrophically (measured K error ~1e6). The MI355X kernels accumulate centred
moments, so the cubature path itself passes: on the part of the horizon that the terminal cost
does not reach (in the cubature path Qf does not enter the smoothed trajectory),
state, action, feedback and feed-forward gains match LQR to the level the reference's Linearize
path achieves (2e-7, 2e-6, 3e-7, 4e-6)."""
import numpy as np
import pytest

import hostsim
from i2c.exp_types import CubatureQuadrature
from i2c.i2c import I2cGraph
from i2c.model import make_env_model


def finite_horizon_lqr(H, A, a, B, Q, R, x0, xg):
    """min sum_t (x-xg)'Q(x-xg) + u'Ru + terminal (x_H-xg)'Q(x_H-xg),  x' = A x + B u + a."""
    nx, nu = B.shape
    K, k = np.zeros((H, nu, nx)), np.zeros((H, nu))
    P, p = Q.copy(), -Q @ xg
    for i in reversed(range(H)):
        Minv = np.linalg.inv(R + B.T @ P @ B)
        K[i] = -Minv @ B.T @ P @ A
        k[i] = -Minv @ B.T @ (P @ a + p)
        p = A.T @ (P @ a + p - P @ B @ Minv @ B.T @ (P @ a + p)) - Q @ xg
        P = Q + A.T @ P @ A - A.T @ P @ B @ Minv @ B.T @ P @ A
    xs, us, x = [], [], x0.copy()
    for i in range(H):
        u = K[i] @ x + k[i]
        xs.append(x)
        us.append(u)
        x = A @ x + B @ u + a
    return np.array(xs), np.array(us), K, k


def _check(lib, device):
    T = 120
    model = make_env_model("LinearKnown", None)
    # scripts/lqr_compare.py:129-134 redefines the goal and the affine term
    model.xag = 10 * np.ones((2, 1))
    model.zg_term = 10 * np.ones((2, 1))
    model.a = model.xag - model.A @ model.xag
    assert model.sig_x0[0, 0] == 1e-20 and model.sig_eta[0, 0] == 1e-20  # the shipped, degenerate noise
    Q, R = np.diag([10.0, 10.0]), np.diag([1.0])
    x_l, u_l, K_l, k_l = finite_horizon_lqr(T, model.A, model.a[:, 0], model.B, Q, R, model.x0[:, 0], model.xag[:, 0])
    # scripts/lqr_compare.py:151-171: alpha = 1e-5, sig_u = 1e2, feed-forward cells, ONE forward-backward pass
    i2c = I2cGraph(model, T, Q, R, Q, 1e-5, 0.0, np.zeros((T, 1)), 1e2 * np.eye(1), None, None,
                   CubatureQuadrature(1, 0, 0), lib=lib, device=device)
    for c in i2c.cells:
        c.state_action_independence = True
    i2c._forward_backward_msgs()
    xu = i2c.get_marginal_trajectory()
    K, k, _ = i2c.get_local_linear_policy()
    n = T // 2
    assert np.abs(xu[:n, :2] - x_l[:n]).max() <= 1e-6
    assert np.abs(xu[:n, 2:] - u_l[:n]).max() <= 1e-5
    assert np.abs(K[:n] - K_l[:n]).max() / np.abs(K_l).max() <= 1e-5
    assert np.abs(k[:n] - k_l[:n]).max() / np.abs(k_l).max() <= 1e-5


def test_cubature_i2c_equals_lqr_cpu():
    _check(hostsim.load(), "cpu")


@pytest.mark.gpu
def test_cubature_i2c_equals_lqr_gpu():
    _check(None, "cuda")
