"""`I2cGraph` / cell views: the reference's solver API (i2c/i2c.py:732-1401) over the MI355X engine.

Callers written against the reference (scripts/i2c_run.py, the covariance-control scripts,
policy/mpc.py) construct `I2cGraph(sys, horizon, Q, R, Qf, alpha, alpha_update_tol, mu_u, sig_u,
mu_x_terminal, sig_x_terminal, inference, res_dir=None)` and read the attributes / methods
listed in SURVEY.md section 8(b). Here every sweep is a HIP kernel launch on device tensors
(`BatchedI2c`); this module only translates shapes and names:

  * reference means are (n, 1) column vectors, covariances (n, n); the engine is (B, T, ...),
  * `graph.cells[t].<name>` is a lazy *view* of cell t of the device buffers,
  * extra keyword arguments (`batch`, `x0`, `sig_x0`, `z_traj`, `device`, `dtype`) expose the
    batch axis the reference does not have; with B == 1 all getters return reference shapes.

`inference` may be `CubatureQuadrature(alpha, beta, kappa)`, `GaussHermiteQuadrature(degree)` or `Linearize()`; not
offered: the matplotlib figures (the `plot_*` methods are no-ops so that runner scripts keep working).
"""
import logging
import os

import numpy as np
import torch

from . import core

BatchedI2c = core.BatchedI2c
from .exp_types import CubatureQuadrature, GaussHermiteQuadrature, Linearize
from .graph_metrics import GraphMetrics

PLOT_TIKZ = False
CHECK_COVAR = False
DEBUG_PLOTS = False


def _np(t):
    return np.array(t.detach().to(torch.float64).cpu().numpy(), copy=True)  # never alias a device/host buffer


class I2cCell:
    """View of one timestep of an `I2cGraph` (reference I2cCell, i2c/i2c.py:51-148).

    Attribute reads fetch the cell's slice from the device buffers; the few attributes the
    reference's callers *write* (`z`, `state_action_independence`, `use_expert_controller`,
    `sys`) are forwarded to the graph."""

    def __init__(self, graph, index):
        object.__setattr__(self, "_g", graph)
        object.__setattr__(self, "index", index)

    # -- helpers --------------------------------------------------------------------------
    def _pick(self, arr, column=False):
        g, t = self._g, self.index
        a = arr[:, t]
        if g.B == 1:
            a = a[0]
            if column:
                a = a.reshape(-1, 1)
        return a

    @property
    def terminal_cell(self):
        return self.index == self._g.H - 1

    @property
    def sys(self):
        return self._g.sys

    @sys.setter
    def sys(self, value):
        self._g.sys = value

    # -- writable flags ---------------------------------------------------------------------
    @property
    def state_action_independence(self):
        e = self._g.engine
        return bool(e.feedforward[(self.index + e.t0) % e.H].item())  # ring row of this cell

    @state_action_independence.setter
    def state_action_independence(self, value):
        e = self._g.engine
        e.feedforward[(self.index + e.t0) % e.H] = 1 if value else 0

    @property
    def use_expert_controller(self):
        return self._g.engine.cell_expert(self.index)

    @use_expert_controller.setter
    def use_expert_controller(self, value):
        # the reference keeps the flag per cell but every script sets all cells alike
        self._g.engine.set_cell_expert(self.index, value)  # per cell, as in the reference (i2c.py:143)

    @property
    def z(self):
        return self._g._cell_target(self.index)

    @z.setter
    def z(self, value):
        self._g._set_cell_target(self.index, value)

    @property
    def sig_xi(self):
        return self._g.sig_xi

    @property
    def temp(self):
        return self._g._maybe_scalar(_np(self._g.engine.temp))

    # -- the cell's own statistics (i2c.py:680-688, 721-730) -----------------------------------
    def _outer_plus(self, mu, sig):
        g = self._g
        z = np.asarray(self.z, float).reshape(-1, g.sys.dim_z)
        d = z - np.asarray(mu, float).reshape(-1, g.sys.dim_z)
        e = d[:, :, None] * d[:, None, :] + np.asarray(sig, float).reshape(-1, g.sys.dim_z, g.sys.dim_z)
        return e[0] if g.B == 1 else e

    def expected_observation_covar(self):
        return self._outer_plus(self.mu_z0_m, self.sig_z0_m)

    def expected_propagated_observation_covar(self):
        return self._outer_plus(self.mu_z0_pf, self.sig_z0_pf)

    def get_obs_covar(self):
        return self._outer_plus(self.mu_z0_m, self.sig_z0_m)

    @staticmethod
    def are_nan(*args):
        return any(np.any(np.isnan(a)) for a in args)

    # -- constants of the problem the reference copies into every cell (i2c.py:54-98) --------------------
    lam_xi = property(lambda self: np.linalg.inv(np.asarray(self._g.sig_xi, float)))
    sig_xi_terminal = property(lambda self: self._g.sig_xi_terminal)
    z_term = property(lambda self: self._g.z_term)
    mu_x_terminal = property(lambda self: self._g.mu_x_terminal)
    sig_x_terminal = property(lambda self: self._g.sig_x_terminal)
    obs_inf = prop_obs_inf = property(lambda self: self._g.obs_inf)

    @property
    def mu_z3_m(self):
        """Terminal observation moments of the smoothed state: the last cell's (i2c.py:565-570), None elsewhere."""
        m = self._g.engine.terminal_observed_marginal()[0] if self.terminal_cell else None
        return None if m is None else (_np(m)[0].reshape(-1, 1) if self._g.B == 1 else _np(m))

    @property
    def sig_z3_m(self):
        s = self._g.engine.terminal_observed_marginal()[1] if self.terminal_cell else None
        return None if s is None else (_np(s)[0] if self._g.B == 1 else _np(s))

    # -- read-only message / posterior state ------------------------------------------------
    def __getattr__(self, name):
        g = object.__getattribute__(self, "_g")
        table = g._cell_table()
        if name not in table:
            raise AttributeError(f"I2cCell has no attribute '{name}' in the MI355X build")
        arr, column = table[name]()
        return self._pick(arr, column)


class I2cGraph(GraphMetrics):
    """Manages Gaussian i2c for a whole trajectory (or a batch of them) on the GPU. (GraphMetrics, graph_metrics.py: the host-side
    bookkeeping surface of the reference's class -- temperature helpers, observation covariances, cost helpers, entropy and
    likelihood metrics.)"""

    def __init__(self, sys, horizon, Q, R, Qf, alpha, alpha_update_tol, mu_u, sig_u, mu_x_terminal,
                 sig_x_terminal, inference, res_dir=None, *, batch=None, x0=None, sig_x0=None, z_traj=None,
                 device=None, dtype=torch.float64, lib=None, group_lanes=0, deterministic_family=False):
        if not isinstance(inference, (CubatureQuadrature, GaussHermiteQuadrature, Linearize)):
            raise ValueError("Unknown inference method")
        if not hasattr(sys, "model_id") or (sys.model_id is None and getattr(sys, "hip_header", None) is None):
            raise TypeError(
                "the MI355X build evaluates models as compiled device functors: `sys` must come from "
                "i2c.model.make_env_model, or be a KnownModel that names the header of its functor (hip_header; "
                "INTEGRATION.md section 3) -- arbitrary Python callables cannot run inside the kernels"
            )
        self.sys = sys
        self.H = int(horizon)
        self.inference = inference
        self.res_dir = res_dir
        self.engine = BatchedI2c(
            sys, horizon, Q, R, Qf, alpha, alpha_update_tol, mu_u, sig_u, mu_x_terminal, sig_x_terminal,
            quad=inference.as_tuple() if isinstance(inference, CubatureQuadrature) else (1.0, 0.0, 0.0),
            x0=x0, sig_x0=sig_x0, z_traj=z_traj, batch=batch, dtype=dtype,
            device=device, lib=lib, keep_zpost=True, keep_prior=True, keep_prior_joint=True,
            inference=("linearize" if isinstance(inference, Linearize) else
                       "gauss_hermite" if isinstance(inference, GaussHermiteQuadrature) else "cubature"),
            gh_degree=getattr(inference, "degree", None), group_lanes=group_lanes, deterministic_family=deterministic_family,  # kernel family (BatchedI2c): 0 = the model's default
        )
        e = self.engine
        self.B = e.B
        self.z = np.copy(np.asarray(sys.zg, dtype=float))
        self.z_term = None if sys.zg_term is None else np.copy(np.asarray(sys.zg_term, dtype=float))
        self.alpha_base = alpha
        self.alpha_update_tol = alpha_update_tol
        self.Q, self.R, self.QR, self.Qf = e.Q, e.R, e.QR, e.Qf
        self.lam_xi0 = np.copy(e.QR)
        self.sig_xi0 = e.sig_xi0
        self.sig_xi_terminal_base = e.sig_xi_terminal_base
        self.mu_x_terminal = None if mu_x_terminal is None else np.asarray(mu_x_terminal, float).reshape(sys.dim_x, 1)
        self.sig_x_terminal = sig_x_terminal
        self.cells = [I2cCell(self, t) for t in range(self.H)]
        self.alpha_risk = []
        self.alpha_sigma = 0
        # the entropy lists _maximize appends to every iteration (i2c.py:1021-1027): LAZY (graph_metrics.py) -- an iteration keeps
        # device-side copies of three covariance rows, the entries are computed when a list is read. On for single-trajectory
        # graphs (the reference's use, its plots), opt-in for batches
        self.record_metrics = self.B == 1
        self.policy_valid = False
        # the engine starts from the constructor's x0 / sig_x0; re-upload only when a caller later CHANGES
        # sys.x0 / sys.sig_x0 (the MPC protocol, mpc.py:149-150)
        self._x0_seen = self._x0_key()
        self.reset_metrics(False)
        self.costs_m_all, self.costs_p_all, self.costs_pf_all = [], [], []
        self._cache = {}

    # ------------------------------------------------------------------ plumbing
    def close(self):
        pass

    def _maybe_scalar(self, arr):
        arr = np.asarray(arr)
        return float(arr.reshape(-1)[0]) if self.B == 1 else arr

    def _x0_key(self):
        return (np.asarray(self.sys.x0, dtype=float).reshape(-1).tobytes(), np.asarray(self.sys.sig_x0, dtype=float).tobytes())

    def _sync_initial_state(self):
        """MPC callers overwrite sys.x0 / sys.sig_x0 between sweeps (mpc.py:149-150)."""
        if self.B != 1:
            return
        x0 = np.asarray(self.sys.x0, dtype=float).reshape(-1)
        s0 = np.asarray(self.sys.sig_x0, dtype=float)
        key = self._x0_key()
        if key != self._x0_seen:
            pack_sym_np = core.engine.pack_sym_np
            e = self.engine
            e.x0.copy_(torch.as_tensor(x0.reshape(-1, 1), dtype=e.dtype))
            e.sig_x0.copy_(torch.as_tensor(pack_sym_np(s0).reshape(-1, 1), dtype=e.dtype))
            self._x0_seen = key

    def _invalidate(self):
        self._cache = {}

    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    def _cell_table(self):
        """name -> thunk returning ((B, T, ...) numpy array, is_column_vector)."""
        e, nx = self.engine, self.engine.nx
        c = self._cached

        def post():
            mu, sig = e.marginal_state_action()
            return _np(mu), _np(sig)

        def pol():
            return tuple(_np(x) for x in e.local_linear_policy())

        def fwd():
            return {k: _np(v) for k, v in e.forward_messages().items()}

        def zp():
            return tuple(_np(x) for x in e.observed_marginal())

        def xm():
            return tuple(_np(x) for x in e.smoothed_next_state())

        def pri():
            return tuple(_np(x) for x in e.prior_state_action())

        def prop():
            return {k: _np(v) for k, v in e.propagated().items()}

        P = lambda: c("post", post)  # noqa: E731
        L = lambda: c("pol", pol)  # noqa: E731
        F = lambda: c("fwd", fwd)  # noqa: E731
        Z = lambda: c("zp", zp)  # noqa: E731
        X = lambda: c("xm", xm)  # noqa: E731
        R_ = lambda: c("pri", pri)  # noqa: E731
        G = lambda: c("prop", prop)  # noqa: E731
        # observation moments of the propagated joint: pushed through sys.observe on the host (GraphMetrics._observed_moments)
        ZP = lambda: c("zpf", lambda: self._observed_moments(G()["mu_xu0_pf"], G()["sig_xu0_pf"]))  # noqa: E731
        return {
            "mu_xu0_m": lambda: (P()[0], True), "sig_xu0_m": lambda: (P()[1], False),
            "mu_xu1_m": lambda: (P()[0], True), "sig_xu1_m": lambda: (P()[1], False),
            "mu_x0_m": lambda: (P()[0][..., :nx], True), "sig_x0_m": lambda: (P()[1][..., :nx, :nx], False),
            "mu_u0_m": lambda: (P()[0][..., nx:], True), "sig_u0_m": lambda: (P()[1][..., nx:, nx:], False),
            "K": lambda: (L()[0], False), "k": lambda: (L()[1], True), "sigK": lambda: (L()[2], False),
            "mu_xu1_f": lambda: (F()["mu_xu1_f"], True), "sig_xu1_f": lambda: (F()["sig_xu1_f"], False),
            "mu_x3_f": lambda: (F()["mu_x3_f"], True), "sig_x3_f": lambda: (F()["sig_x3_f"], False),
            "J_dyn": lambda: (F()["J_dyn"], False), "Jx_dyn": lambda: (F()["J_dyn"][..., :nx, :], False),
            "mu_z0_m": lambda: (Z()[0], True), "sig_z0_m": lambda: (Z()[1], False),
            "mu_x3_m": lambda: (X()[0], True), "sig_x3_m": lambda: (X()[1], False),
            "mu_xu0_f": lambda: (R_()[0], True), "sig_xu0_f": lambda: (R_()[1], False),
            "mu_u0_f": lambda: (R_()[0][..., nx:], True), "sig_u0_f": lambda: (R_()[1][..., nx:, nx:], False),
            "mu_xu0_pf": lambda: (G()["mu_xu0_pf"], True), "sig_xu0_pf": lambda: (G()["sig_xu0_pf"], False),
            "mu_x0_pf": lambda: (G()["mu_xu0_pf"][..., :nx], True),
            "sig_x0_pf": lambda: (G()["sig_xu0_pf"][..., :nx, :nx], False),
            "mu_u0_pf": lambda: (G()["mu_xu0_pf"][..., nx:], True),
            "sig_u0_pf": lambda: (G()["sig_xu0_pf"][..., nx:, nx:], False),
            "mu_x3_pf": lambda: (G()["mu_x3_pf"], True), "sig_x3_pf": lambda: (G()["sig_x3_pf"], False),
            "mu_z0_pf": lambda: (ZP()[0], True), "sig_z0_pf": lambda: (ZP()[1], False),
            # derived views the reference keeps as cell attributes: the state message entering the cell (x0 for cell 0, the previous
            # cell's prediction after it), slices of the updated joint, the observation moments of the PRIOR joint (pushed through
            # sys.observe on the host), the smoother's lag covariance J_x sig_x3_m (i2c.py:578), constants of the problem
            "mu_x0_f": lambda: (self._entering_state()[0], True), "sig_x0_f": lambda: (self._entering_state()[1], False),
            "mu_x1_f": lambda: (F()["mu_xu1_f"][..., :nx], True), "sig_x1_f": lambda: (F()["sig_xu1_f"][..., :nx, :nx], False),
            "mu_u1_f": lambda: (F()["mu_xu1_f"][..., nx:], True), "sig_u1_f": lambda: (F()["sig_xu1_f"][..., nx:, nx:], False),
            "mu_z0_f": lambda: (c("zf", self._prior_observation)[0], True), "sig_z0_f": lambda: (c("zf", self._prior_observation)[1], False),
            "sig_x_lag_m": lambda: (np.einsum("btij,btjk->btik", F()["J_dyn"][..., :nx, :], X()[1]), False),
            "sig_eta": lambda: (self._const_cells(self.sys.sig_eta), False), "sig_eta_pf": lambda: (self._const_cells(self.sys.sig_eta), False),
            "mu_u0_base": lambda: (np.array(e.mu_u0_base), True), "sig_u0_base": lambda: (self._const_cells(e.sig_u0_base), False),
            # Riccati-form backward messages (after _backward_ricatti_msgs, i2c.py:612-678)
            "nu_x0_b": lambda: (self._riccati_np()[0], True), "lambda_x0_b": lambda: (self._riccati_np()[1], False),
            "nu_x3_b": lambda: (self._riccati_np()[2], True), "lambda_x3_b": lambda: (self._riccati_np()[3], False),
        }

    def _const_cells(self, a):
        a = np.asarray(a, float)
        return np.broadcast_to(a, (self.B, self.H) + a.shape)

    def _prior_observation(self):
        """(mu_z0_f, sig_z0_f): the prior joint of the last forward sweep through sys.observe, plus the cost noise alpha xi at the
        temperature that sweep ran at -- the innovation covariance, as the reference stores it (i2c.py:390-393, 402)."""
        mz, sz = self._observed_moments(*(_np(x) for x in self.engine.prior_state_action()))
        a = _np(getattr(self.engine, "alpha_fwd", self.engine.alpha)).reshape(self.B, 1, 1, 1)
        return mz, sz + a * np.asarray(self.sig_xi0, float)

    def _entering_state(self):
        """(mu_x0_f, sig_x0_f) of every cell as (B, T, nx), (B, T, nx, nx): the belief over the initial state for cell 0, the
        previous cell's prediction afterwards (i2c.py:876-880)."""
        def make():
            f = {k: _np(v) for k, v in self.engine.forward_messages().items()}
            m0 = _np(self.engine.x0).T.reshape(self.B, 1, -1)
            s0 = _np(self.engine._sym_rows(self.engine.sig_x0.unsqueeze(0), 0, self.engine.nx))[:, :1]
            return np.concatenate((m0, f["mu_x3_f"][:, :-1]), axis=1), np.concatenate((s0, f["sig_x3_f"][:, :-1]), axis=1)

        return self._cached("x0f", make)

    def _riccati_np(self):
        """(nu_x0_b, lambda_x0_b, nu_x3_b, lambda_x3_b) as (B, T, ...) arrays. The message entering cell t at x3 is the
        one leaving cell t+1 at x0; at the end of the chain it is posterior minus filter in information form
        (i2c.py:615-617)."""
        if getattr(self, "_riccati", None) is None:
            raise AttributeError("call _backward_ricatti_msgs() first (i2c.py:888-893)")

        def make():
            nu0, lam0 = (_np(x) for x in self._riccati)
            f = {k: _np(v) for k, v in self.engine.forward_messages().items()}
            m3m, s3m = (_np(x) for x in self.engine.smoothed_next_state())
            lam_m, lam_f = np.linalg.inv(s3m[:, -1]), np.linalg.inv(f["sig_x3_f"][:, -1])
            nu_end = (np.einsum("bij,bj->bi", lam_m, m3m[:, -1]) - np.einsum("bij,bj->bi", lam_f, f["mu_x3_f"][:, -1]))
            nu3 = np.concatenate((nu0[:, 1:], nu_end[:, None]), axis=1)
            lam3 = np.concatenate((lam0[:, 1:], (lam_m - lam_f)[:, None]), axis=1)
            return nu0, lam0, nu3, lam3

        return self._cached("riccati", make)

    def _cell_target(self, t):
        e = self.engine
        if e.z is None:
            return np.copy(self.z)
        z = _np(e.z[(t + e.t0) % e.H]).T  # (B, nz); ring row of cell t
        return z[0].reshape(-1, 1) if self.B == 1 else z

    def _set_cell_target(self, t, value):
        """Per-cell targets (MPC reference trajectories, mpc.py:29-31)."""
        e = self.engine
        if e.z is None:
            zg = torch.as_tensor(e.zg, dtype=e.dtype, device=e.device)
            e.z = zg.reshape(1, -1, 1).repeat(self.H, 1, self.B).contiguous()
            e.refresh_problem()
        v = torch.as_tensor(np.asarray(value, dtype=float).reshape(-1, e.nz).T, dtype=e.dtype, device=e.device)
        e.z[(t + e.t0) % e.H] = v.expand(e.nz, self.B)

    # ------------------------------------------------------------------ temperature
    @property
    def alpha(self):
        return self._maybe_scalar(_np(self.engine.alpha))

    @alpha.setter
    def alpha(self, value):
        self.engine.alpha.copy_(torch.as_tensor(np.broadcast_to(np.asarray(value, float), (self.B,)).copy(),
                                                dtype=self.engine.dtype))
        self.engine._broadcast_alpha()

    @property
    def sig_xi(self):
        a = _np(self.engine.alpha)
        return a[0] * self.sig_xi0 if self.B == 1 else a[:, None, None] * self.sig_xi0

    @property
    def sig_xi_terminal(self):
        if self.sig_xi_terminal_base is None:
            return None
        a = _np(self.engine.alpha)
        return a[0] * self.sig_xi_terminal_base if self.B == 1 else a[:, None, None] * self.sig_xi_terminal_base

    @property
    def tau(self):
        return self.engine.tau

    @tau.setter
    def tau(self, value):
        self.engine.tau = int(value)

    @property
    def _propagate(self):
        return self.engine._propagate

    @_propagate.setter
    def _propagate(self, value):
        self.engine._propagate = bool(value)

    @property
    def state_action_independence(self):
        return bool(self.engine.feedforward.all().item())

    @state_action_independence.setter
    def state_action_independence(self, value):  # mpc.py:21,37,40 set this graph-level attribute
        pass

    def _hist(self, lst):
        h = BatchedI2c.history(lst)
        return [self._maybe_scalar(row) for row in h]

    alphas = property(lambda self: self._hist(self.engine.alphas))
    alphas_desired = property(lambda self: self._hist(self.engine.alphas_desired))
    alphas_pf = property(lambda self: self._hist(self.engine.alphas_pf))
    costs_m = property(lambda self: self._hist(self.engine.costs_m))
    costs_m_var = property(lambda self: self._hist(self.engine.costs_m_var))
    costs_pf = property(lambda self: self._hist(self.engine.costs_pf))
    costs_pf_var = property(lambda self: self._hist(self.engine.costs_pf_var))
    kl_terms = property(lambda self: self._hist(self.engine.kl_terms))

    @property
    def em_iter(self):
        return self.engine.em_iter

    @em_iter.setter
    def em_iter(self, v):
        self.engine.em_iter = int(v)

    # ------------------------------------------------------------------ sweeps / EM
    def _check(self):
        if self.B == 1:
            self.engine.raise_on_failure()

    def _forward_msgs(self):
        self._sync_initial_state()
        self._invalidate()
        self.engine.forward_sweep()
        self._check()

    def _backward_msgs(self):
        self._invalidate()
        self.engine.backward_sweep()
        self._check()

    def _forward_backward_msgs(self):
        self._forward_msgs()
        self._backward_msgs()

    def _backward_ricatti_msgs(self):
        """i2c.py:888-893 (the reference's spelling). Cells then expose the Riccati-form K / k / sigK and
        nu_x0_b / lambda_x0_b."""
        self._invalidate()
        self._riccati = self.engine.riccati_sweep()
        self._check()
        self.policy_valid = True

    def _update_priors(self):
        self.engine.update_priors()

    def propagate(self):
        self._sync_initial_state()
        self._invalidate()
        self.engine.propagate()
        self._check()

    def calibrate_alpha(self, only_decrease=False):
        self._sync_initial_state()
        self._invalidate()
        before = self.alpha
        self.engine.calibrate_alpha(only_decrease)
        logging.info(f"calibrating alpha from propagation {before}->{self.alpha}")

    def _check_iteration(self):
        """After an M-step, single trajectory: failure status AND alpha_hat with one synchronisation (engine.iteration_health)."""
        if self.B != 1:
            return
        status, alpha_hat = self.engine.iteration_health()
        self.engine.raise_on_failure(status)
        if np.isnan(alpha_hat).any():
            raise ValueError("Alpha is NaN")

    def _checked_iteration(self):
        """The tail of an iteration: metric snapshots (device-side copies, i2c.py:1021-1027) enqueued BEFORE the one synchronisation
        of _check_iteration() -- they run while the host waits instead of after it -- and taken back if the iteration raises."""
        if self.record_metrics:
            self._append_iteration_metrics()
        try:
            self._check_iteration()
        except Exception:
            if self.record_metrics:
                self._drop_last_iteration_metrics()
            raise

    def _maximize(self):
        self.engine.maximize(update_alpha=True)
        self._invalidate()
        self._checked_iteration()

    def compute_update_alpha(self, update_alpha):
        """i2c.py:921-963: alpha_hat, the clamp, and (with update_alpha) the new temperature in every cell -- nothing else:
        no cost bookkeeping, no _update_priors, no KL term (those belong to _maximize, i2c.py:1004-1019)."""
        self.engine.alpha_mstep(update_alpha=bool(update_alpha))

    def learn_msgs(self):
        self._sync_initial_state()
        self._invalidate()
        self.engine.learn_msgs()
        self._checked_iteration()

    def update_models(self):
        pass

    # ------------------------------------------------------------------ getters (i2c.py:1191-1314)
    def _squeeze(self, a):
        return a[0] if self.B == 1 else a

    def get_local_linear_policy(self):
        K, k, sigK = (_np(x) for x in self.engine.local_linear_policy())
        return self._squeeze(K), self._squeeze(k), self._squeeze(sigK)

    def get_local_expert_linear_policy(self):
        K, _, sigK = (_np(x) for x in self.engine.local_linear_policy())
        mu, sig = (_np(x) for x in self.engine.marginal_state_action())
        nx = self.engine.nx
        lam = np.linalg.inv(sig[..., :nx, :nx])
        return tuple(self._squeeze(a) for a in (K, mu[..., nx:], sigK, mu[..., :nx], lam))

    def get_marginal_input(self):
        mu = _np(self.engine.marginal_state_action()[0])[..., self.engine.nx:]
        return mu[0][:, :, None] if self.B == 1 else mu

    def get_marginal_state_action(self):
        mu = _np(self.engine.marginal_state_action()[0])
        return mu[0][:, :, None] if self.B == 1 else mu

    def get_marginal_state_action_distribution(self):
        mu, sig = (_np(x) for x in self.engine.marginal_state_action())
        return self._squeeze(mu), self._squeeze(sig)

    def get_marginal_trajectory(self):
        return self._squeeze(_np(self.engine.marginal_state_action()[0]))

    def get_marginal_observed_trajectory(self):
        mz = self._squeeze(_np(self.engine.observed_marginal()[0]))
        mzt, _ = self.engine.terminal_observed_marginal()
        if mzt is None:
            return mz, None
        mzt = _np(mzt)
        return mz, (mzt[0].reshape(1, -1) if self.B == 1 else mzt)

    @staticmethod
    def list_minima(list_, n_min, n_steps):
        """i2c.py:1172-1182."""
        if len(list_) > n_min:
            if len(list_) > n_steps and n_steps > 0:
                return all(list_[-1 - i] < list_[-2 - i] for i in range(n_steps))
            return False

    @staticmethod
    def indexed_confidence_bound(mu, sig, idx):
        """Two-sigma band of component idx along a trajectory (i2c.py:1184-1189)."""
        std = 2.0 * np.sqrt(sig[:, idx, idx])
        return mu[:, idx] + std, mu[:, idx] - std

    def get_state_action_prior(self):
        mu = _np(self.engine.prior_state_action()[0])
        return mu[0][:, :, None] if self.B == 1 else mu

    def get_state_and_action(self):
        mu = _np(self.engine.marginal_state_action()[0])
        nx = self.engine.nx
        if self.B == 1:
            return mu[0][:, :nx, None], mu[0][:, nx:, None]
        return mu[..., :nx], mu[..., nx:]

    def get_propagated_state_action(self):
        p = self.engine.propagated()
        return self._squeeze(_np(p["mu_xu0_pf"])), self._squeeze(_np(p["sig_xu0_pf"]))

    def get_propagated_state(self):
        p = self.engine.propagated()
        nx = self.engine.nx
        mu, sig = _np(p["mu_xu0_pf"])[..., :nx], _np(p["sig_xu0_pf"])[..., :nx, :nx]
        if self.B == 1:
            return mu[0][:, :, None], sig[0]
        return mu, sig

    def get_z_covar(self):
        mz, sz = (_np(x) for x in self.engine.observed_marginal())
        err = self._targets() - mz
        return self._squeeze((err[..., :, None] * err[..., None, :] + sz).sum(axis=1))

    def converged(self):
        c = self.costs_m
        if len(c) > 2 and self.B == 1:
            return abs(c[-1] - c[-2]) / c[-1] < 0.005
        return False

    # ------------------------------------------------------------------ metrics / persistence
    def reset_metrics(self, extend=True):
        e = self.engine
        if extend:
            self.costs_m_all.extend(self.costs_m)
            self.costs_pf_all.extend(self.costs_pf)
        e.costs_m, e.costs_m_var, e.costs_pf, e.costs_pf_var, e.kl_terms = [], [], [], [], []
        e.em_iter = 0
        self._pending_metrics = []  # (snapshots of iterations whose entropy entries nobody has read yet, graph_metrics.py)
        for name in ("likelihoods", "likelihoods_xu", "likelihoods_z", "policy_entropy", "sig_eta_entropy",
                     "sig_eta_pf_entropy", "x_prior_entropy", "x_prior_neg_entropy", "propagate_entropy",
                     "costs_p", "costs_s", "risk"):
            setattr(self, name, [])

    def reset_priors(self):
        raise NotImplementedError("reset_priors: construct a new I2cGraph instead")

    def save_traj(self, res_dir):
        """Same four .npy files as the reference (i2c.py:1374-1382)."""
        mu = _np(self.engine.marginal_state_action()[0])
        mz = _np(self.engine.observed_marginal()[0])
        nx = self.engine.nx
        sq = (lambda a: a[0][:, :, None]) if self.B == 1 else (lambda a: a)
        np.save(os.path.join(res_dir, "xu_plan.npy"), np.hstack((sq(mu[..., :nx]), sq(mu[..., nx:]))) if self.B == 1 else mu)
        np.save(os.path.join(res_dir, "x_plan.npy"), sq(mu[..., :nx]))
        np.save(os.path.join(res_dir, "u_plan.npy"), sq(mu[..., nx:]))
        np.save(os.path.join(res_dir, "z_plan.npy"), sq(mz))

    _STATE_TENSORS = ("post", "alpha", "temp", "feedforward", "status", "x0", "sig_x0", "cell_init")
    _STATE_OPTIONAL = ("z", "alpha_cell", "alpha_init", "expert_cells")
    _STATE_LISTS = ("alphas", "alphas_desired", "alphas_pf", "costs_m", "costs_m_var", "costs_pf", "costs_pf_var", "kl_terms")

    def state_dict(self):
        """Everything a resumed EM or MPC run needs (replaces the reference's whole-object dill pickle, i2c.py:1384-1398):
        posterior / prior state, temperatures (per trajectory and per cell), the belief x0 / sig_x0, per-cell targets,
        mode flags, the moving terminal-cell index, tau, and the metric histories."""
        e = self.engine
        sd = {k: getattr(e, k).detach().cpu().clone() for k in self._STATE_TENSORS}
        sd["prior"] = e.prior.detach().cpu().clone()
        for k in self._STATE_OPTIONAL:
            v = getattr(e, k)
            sd[k] = None if v is None else v.detach().cpu().clone()
        for k in self._STATE_LISTS:
            sd[k] = [t.detach().cpu().clone() for t in getattr(e, k)]
        sd.update(em_iter=e.em_iter, tau=e.tau, terminal_cell=e.terminal_cell, t0=e.t0, propagate=bool(e._propagate),
                  use_expert_controller=bool(e.use_expert_controller), horizon=e.H, batch=e.B, model=self.sys.model_name)
        return sd

    def load_state_dict(self, sd):
        e = self.engine
        if (sd["horizon"], sd["batch"], sd["model"]) != (e.H, e.B, self.sys.model_name):
            raise ValueError("checkpoint is for another problem (model / horizon / batch)")
        for k in self._STATE_TENSORS:
            getattr(e, k).copy_(sd[k])
        if e.prior is not e.post:
            e._post_spare, e.prior = e.prior, e.post
        if not torch.equal(sd["prior"], sd["post"]):  # saved between a sweep and its _update_priors()
            e.post, e._post_spare = e._post_spare, None
            e.post.copy_(sd["post"])
            e.prior.copy_(sd["prior"])
        if sd["alpha_cell"] is not None:
            e.enable_per_cell_alpha()
        if sd.get("expert_cells") is not None and e.expert_cells is None:
            e.set_cell_expert(0, e.use_expert_controller)  # allocates the per-cell flags (and hands them to the library)
        if sd["z"] is not None and e.z is None:
            e.set_targets(np.transpose(sd["z"].double().numpy(), (2, 0, 1)))
        for k in self._STATE_OPTIONAL:
            if sd.get(k) is not None:
                getattr(e, k).copy_(sd[k])
        for k in self._STATE_LISTS:
            setattr(e, k, [t.to(e.device) for t in sd[k]])
        e.em_iter, e.tau, e.terminal_cell, e.t0 = int(sd["em_iter"]), int(sd["tau"]), int(sd["terminal_cell"]), int(sd["t0"])
        e._propagate, e.use_expert_controller = bool(sd["propagate"]), bool(sd["use_expert_controller"])
        e.refresh_problem()
        self._invalidate()

    def save(self, path, name):
        """i2c.py:1384-1390 (`i2c_{name}.pkl` there, a dill pickle of the whole graph): the solver state as tensors."""
        torch.save(self.state_dict(), os.path.join(path, f"i2c_{name}.pt"))

    @classmethod
    def load(cls, path, *args, **kwargs):
        """Counterpart of the reference's I2cGraph.load (i2c.py:1392-1398). A dill pickle carries the constructor
        arguments along; a tensor checkpoint does not, so they are passed again: I2cGraph.load(path, sys, horizon, ...)."""
        g = cls(*args, **kwargs)
        g.load_state_dict(torch.load(path, weights_only=False))
        return g

    # figures are presentation, not part of the solver: keep runner scripts working
    def _no_plot(self, *args, **kwargs):
        return None

    plot_traj = plot_metrics = plot_alphas = plot_cost = plot_controller = plot_observed_traj = _no_plot
    plot_propagate = plot_uncertainty = plot_cost_all = plot_system_dynamics = plot_ricatti = plot_terminal_observed_traj = _no_plot
