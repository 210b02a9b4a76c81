"""Helpers the reference's runner scripts import from i2c.utils: the closed-form LQR solver used as a
known answer, the rollout cost evaluators, and results-folder / logging / seeding utilities.
Same names and call signatures as the reference (i2c/utils.py); plots are no-ops."""
import datetime
import logging
import os

import numpy as np

DATETIME = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")


def quadratic_trajectory_cost(z, z_term, zg, zg_term, QR, Qf):
    e, et = z - zg.reshape(1, -1), z_term - zg_term.reshape(1, -1)
    return float(np.einsum("bi,ij,bj->", e, QR, e) + (et @ Qf @ et.T).item())


def finite_horizon_lqr(H, A, a, B, Q, R, x0, xg, ug, dim_x, dim_u):
    """Affine finite-horizon LQR by backward Riccati recursion; returns (x, u, K, k, cost, P_t, p_t)."""
    K, k = np.zeros((H, dim_u, dim_x)), np.zeros((H, dim_u))
    Ps, ps = np.zeros((H, dim_x, dim_x)), np.zeros((H, dim_x))
    P, p = np.array(Q, dtype=float), -Q @ xg
    for i in reversed(range(H)):
        Ps[i], ps[i] = P, p
        Minv = np.linalg.inv(R + B.T @ P @ B)
        rhs = B.T @ (P @ a + p) - R @ ug
        K[i] = -Minv @ B.T @ P @ A
        k[i] = -Minv @ rhs
        p = A.T @ (P @ a + p - P @ B @ Minv @ rhs) - Q @ xg
        P = Q + A.T @ P @ A - A.T @ P @ B @ Minv @ B.T @ P @ A
    xs, us, x, cost = np.zeros((H, dim_x)), np.zeros((H, dim_u)), np.array(x0, dtype=float), 0.0
    for i in range(H):
        u = K[i] @ x + k[i]
        xs[i], us[i] = x, u
        cost += (x - xg) @ Q @ (x - xg) + (u - ug) @ R @ (u - ug)
        x = A @ x + B @ u + a
    cost += (x - xg) @ Q @ (x - xg)
    return xs, us, K, k, cost, Ps, ps


def finite_horizon_lqr_tv(H, A, a, B, Q, R, Qf, q, r, qf, x0, dim_x, dim_u):
    """Time-varying affine LQR with linear cost terms: feedback gains K (H, nu, nx) and offsets k (H, nu) of u_t = K_t x_t + k_t for
    x' = A_t x + B_t u + a_t and the stage cost x^T Q_t x - 2 q_t^T x + u^T R_t u - 2 r_t^T u (terminal: Qf, qf); the reference's
    signature and sign conventions (utils.py:30-56). A, B, Q, R, a, q, r are stacked over the horizon, vectors as columns."""
    K, k = np.zeros((H, dim_u, dim_x)), np.zeros((H, dim_u))
    P, p = np.asarray(Qf, float), -np.asarray(qf, float).reshape(dim_x, 1)
    for i in reversed(range(H)):
        Ai, Bi = np.asarray(A[i], float), np.asarray(B[i], float)
        ai, qi, ri = (np.asarray(v[i], float).reshape(-1, 1) for v in (a, q, r))
        Minv = np.linalg.inv(np.asarray(R[i], float) + Bi.T @ P @ Bi)
        drift = P @ ai + p
        rhs = Bi.T @ drift - ri
        K[i] = -Minv @ Bi.T @ P @ Ai
        k[i] = (-Minv @ rhs).reshape(-1)
        p = Ai.T @ (drift - P @ Bi @ Minv @ rhs) - qi
        P = np.asarray(Q[i], float) + Ai.T @ P @ Ai - Ai.T @ P @ Bi @ Minv @ Bi.T @ P @ Ai
    return K, k


class _Evaluator:
    def __init__(self, W, Wf, sg, sg_term, dim_x):
        self.W, self.Wf = np.asarray(W, float), np.asarray(Wf, float)
        self.sg, self.sg_term = np.reshape(sg, (-1, 1)), np.reshape(sg_term, (-1, 1))
        self.dim_x = dim_x
        self.dim_s = self.W.shape[0]
        assert self.W.shape[1] == self.dim_s and self.sg.shape[0] == self.dim_s
        self.planned_cost = []

    def _eval_traj(self, s, s_term):
        """Quadratic cost of one rollout; like the reference it leaves out the last running step
        (utils.py:167-168) and adds the terminal term on the first dim_x terminal features."""
        err = np.asarray(s)[:-1] - self.sg.reshape(1, -1)
        cost = float(np.einsum("bi,ij,bj->", err, self.W, err))
        if s_term is not None:
            et = (np.reshape(s_term, (1, -1)) - self.sg_term.reshape(1, -1))[-1, : self.dim_x]
            cost += float(et @ self.Wf @ et)
        return cost

    def plot(self, *a, **k):
        return None

    plot_sample = plot


class TrajectoryEvaluator(_Evaluator):
    def __init__(self, W, Wf, sg, sg_term, dim_x):
        super().__init__(W, Wf, sg, sg_term, dim_x)
        self.actual_cost = []

    def eval(self, actual_traj, actual_terminal, planned_traj, planned_terminal):
        self.actual_cost.append(self._eval_traj(actual_traj, actual_terminal))
        self.planned_cost.append(self._eval_traj(planned_traj, planned_terminal))

    def save(self, name, res_dir):
        np.save(os.path.join(res_dir, f"cost_actual_{name}.npy"), np.asarray(self.actual_cost))
        np.save(os.path.join(res_dir, f"cost_plan_{name}.npy"), np.asarray(self.planned_cost))


class StochasticTrajectoryEvaluator(_Evaluator):
    """Mean / min / max / 10th / 90th percentile of the rollout costs per evaluation (utils.py:150-265)."""

    def __init__(self, W, Wf, sg, sg_term, dim_x):
        super().__init__(W, Wf, sg, sg_term, dim_x)
        self.mu_actual_cost, self.max_actual_cost, self.min_actual_cost = [], [], []
        self.actual_cost_10, self.actual_cost_90 = [], []

    def eval(self, actual_trajs, actual_trajs_term, planned_traj, planned_traj_term):
        costs = np.array([self._eval_traj(s, st) for s, st in zip(actual_trajs, actual_trajs_term)])
        self.mu_actual_cost.append(costs.mean())
        self.min_actual_cost.append(costs.min())
        self.max_actual_cost.append(costs.max())
        p10, p90 = np.percentile(costs, (10, 90))
        self.actual_cost_10.append(p10)
        self.actual_cost_90.append(p90)
        self.planned_cost.append(self._eval_traj(planned_traj, planned_traj_term))

    def save(self, name, res_dir):
        np.save(os.path.join(res_dir, f"cost_actual_mean_{name}.npy"), np.asarray(self.mu_actual_cost))
        np.save(os.path.join(res_dir, f"cost_plan_{name}.npy"), np.asarray(self.planned_cost))


def set_seed(seed):
    np.random.seed(int(seed))  # the reference passes argparse's str through


def make_results_folder(config, seed, name, folder_name="_results", release=False):
    parts = [config.replace(" ", "-"), str(seed), name.replace(" ", "-")]
    res_dir = os.path.join(folder_name, "_".join((["release"] if release else [DATETIME]) + parts))
    os.makedirs(res_dir, exist_ok=True)
    return res_dir


def configure_plots():
    pass


def setup_logger(res_dir, level=logging.INFO):
    for h in logging.root.handlers[:]:
        logging.root.removeHandler(h)
    logging.basicConfig(filename=os.path.join(res_dir, "output.log"), level=level,
                        format="[%(asctime)s] %(pathname)s:%(lineno)d %(levelname)s- %(message)s")


def write_commit(res_dir):
    try:
        import subprocess

        sha = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        sha = "unknown"
    with open(os.path.join(res_dir, "git_commit.txt"), "w") as f:
        f.write(sha)


# ---- figure helpers the covariance-control scripts import (reference i2c/utils.py:378-419) ----
def covariance_2d(covar, mean, axis, n_std=2.0, facecolor="b", **kwargs):
    """Outline of the n_std ellipse of a 2-D Gaussian on `axis` (scripts/nonlinear_covariance_control.py:32-52,
    scripts/linear_gaussian_covariance_control.py:38-55). Returns the patch."""
    from matplotlib.patches import Ellipse

    covar = np.asarray(covar, dtype=float)
    evals, evecs = np.linalg.eigh(0.5 * (covar + covar.T))
    if not np.all(np.isfinite(evals)) or np.any(evals < 0.0):
        raise ValueError("covariance_2d needs a finite positive semi-definite 2x2 covariance")
    major = evecs[:, 1]  # eigh sorts ascending: column 1 is the long axis
    patch = Ellipse(
        xy=np.asarray(mean, dtype=float).reshape(-1)[:2],
        width=2.0 * n_std * float(np.sqrt(evals[1])),
        height=2.0 * n_std * float(np.sqrt(evals[0])),
        angle=float(np.degrees(np.arctan2(major[1], major[0]))),
        edgecolor=facecolor,
        facecolor="none",
        **kwargs,
    )
    return axis.add_patch(patch)


def plot_uncertainty(ax, x, mean, variance, stds=(2,), color="b", alpha=0.1):
    """Shaded +/- k sigma bands around a mean curve."""
    x, mean, variance = (np.asarray(v, dtype=float).squeeze() for v in (x, mean, variance))
    for k in stds:
        half = k * np.sqrt(variance)
        ax.fill_between(x, mean + half, mean - half, where=half > 0, color=color, alpha=alpha)
