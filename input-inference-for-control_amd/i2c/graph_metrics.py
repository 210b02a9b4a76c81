"""Host-side bookkeeping surface of the reference's `I2cGraph` that is neither a message nor a controller
(i2c/i2c.py:913-992, 1029-1229, 1294-1298): the temperature helpers, observation covariances, cost helpers, the entropy /
likelihood metrics that `_maximize` appends to every iteration, the KL divergence.

Everything here is NumPy on arrays the graph already exposes (posterior, prior, forward messages, propagation) plus -- where the
reference pushes a Gaussian through `sys.observe` / `sys.forward` on the host (compute_cost_gaussian, the propagated observation
covariance, the likelihood) -- the host-side `QuadratureInference` of this package with the graph's rule. Not the solver: the EM
iteration itself (costs, alpha_hat, KL term of covariance control) is computed by the kernels (`BatchedI2c.maximize`).
Batched graphs (`batch=B`) get one value per trajectory (leading axis B); a single-trajectory graph gets the reference's scalars
and `(n, 1)` / `(n, n)` shapes."""
import warnings

import numpy as np

from . import core
from .exp_types import CubatureQuadrature, GaussHermiteQuadrature
from .inference.quadrature import QuadratureInference

sym_size, unpack_sym = core.engine.sym_size, core.engine.unpack_sym

_TWO_PI_E = 2.0 * np.pi * np.e


def _sum_gaussian_entropy(sig, what):
    """sum over the horizon of 0.5 log det(2 pi e sig_t); sig: (B, T, n, n) -> (B,). A covariance that is not positive definite gives
    nan with a warning, as the reference's 0.5 log(det(.)) does (i2c.py:1072-1133): a metric must not abort an EM run."""
    sign, logdet = np.linalg.slogdet(_TWO_PI_E * np.asarray(sig, float))
    if np.any(sign <= 0):
        b, t = np.argwhere(sign <= 0)[0]
        warnings.warn(f"{what}: cell {t} (trajectory {b}) has a covariance that is not positive definite: entropy = nan", RuntimeWarning)
        logdet = np.where(sign > 0, logdet, np.nan)
    return 0.5 * logdet.sum(axis=1)


# the lists _maximize appends an entropy to in every EM iteration (i2c.py:1021-1027). Here they are LAZY: an iteration only keeps
# device-side copies of the three covariance rows the entropies are functions of (no device -> host copy, no synchronisation inside
# the EM loop); the values are computed when a list is first READ.
_LAZY_LISTS = ("policy_entropy", "sig_eta_entropy", "sig_eta_pf_entropy", "x_prior_entropy", "x_prior_neg_entropy", "propagate_entropy")
_MAX_PENDING_BYTES = 256 << 20  # device memory the pending snapshots may hold before they are folded into the lists anyway


def _lazy_list(name):
    def get(self):
        self._materialise_metrics()
        return self.__dict__.setdefault("_m_" + name, [])

    def set_(self, value):
        self.__dict__["_m_" + name] = value

    return property(get, set_, doc=f"{name}: one entry per EM iteration, computed on first read (i2c.py:1021-1027)")


class GraphMetrics:
    """Mixin of I2cGraph (graph.py). Uses: self.engine, self.sys, self.B, self.H, self.QR, self.Qf, self.z, self.z_term,
    self._cell_table(), self._cell_target(t), self._maybe_scalar(), self._squeeze(), the metric lists of reset_metrics()."""

    policy_entropy = _lazy_list("policy_entropy")
    sig_eta_entropy = _lazy_list("sig_eta_entropy")
    sig_eta_pf_entropy = _lazy_list("sig_eta_pf_entropy")
    x_prior_entropy = _lazy_list("x_prior_entropy")
    x_prior_neg_entropy = _lazy_list("x_prior_neg_entropy")
    propagate_entropy = _lazy_list("propagate_entropy")

    # ------------------------------------------------------------------ host push-through with the graph's rule
    @property
    def obs_inf(self):
        """The graph-level observation transform of the reference (i2c.py:839-846): the graph's rule, the unit cubature rule
        under Linearize()."""
        if getattr(self, "_obs_inf", None) is None:
            rule = self.inference if isinstance(self.inference, (CubatureQuadrature, GaussHermiteQuadrature)) else CubatureQuadrature(1, 0, 0)
            self._obs_inf = QuadratureInference(rule, self.sys.dim_xu)
        return self._obs_inf

    def _table(self, name):
        return self._cell_table()[name]()[0]  # (B, T, ...) numpy

    def _targets(self):
        """(B, T, nz): the target of every cell (one shared by the batch unless per-cell targets were set)."""
        nz = self.sys.dim_z
        return np.stack([np.broadcast_to(np.asarray(self._cell_target(t), float).reshape(-1, nz), (self.B, nz)) for t in range(self.H)], axis=1)

    def _per_traj(self, values):
        v = np.asarray(values)
        return v[0] if self.B == 1 else v

    # ------------------------------------------------------------------ temperature (i2c.py:913-981)
    def calculate_alpha(self, z_covar, z_covar_term=None):
        """alpha_hat = tr(QR E[(z - z*)(z - z*)^T]) / (dim_z H), with the terminal term when given (i2c.py:913-919)."""
        z_covar = np.asarray(z_covar, float)
        tr = np.trace(self.QR @ z_covar, axis1=-2, axis2=-1)
        n = float(self.sys.dim_z * self.H)
        if z_covar_term is not None:
            tr = tr + np.trace(self.Qf @ np.asarray(z_covar_term, float), axis1=-2, axis2=-1)
            n += float(self.sys.dim_z_term)
        return tr / n

    def update_alpha(self, alpha_update):
        """The clamp of the M-step (i2c.py:948-963): the new temperature stays within [tol, 2 - tol] x the old one; a negative
        tolerance freezes it."""
        new = np.asarray(alpha_update, float)
        if np.any(np.isnan(new)):
            raise ValueError("Alpha is NaN")
        old, tol = np.asarray(self.alpha, float), self.alpha_update_tol
        if tol >= 0.0:
            self.alpha_update_tol_u = 2.0 - tol
            new = np.clip(new, tol * old, self.alpha_update_tol_u * old)
        else:
            new = old
        self._update_alpha(new)

    def _update_alpha(self, update):
        self.alpha = update  # (the cells read the graph's temperature: sig_xi = alpha sig_xi0 everywhere)

    def _override_alpha(self, update):
        e = self.engine
        if e.alphas:
            e.alphas[-1] = e.alpha.new_tensor(np.broadcast_to(np.asarray(update, float), (self.B,)).copy())
        self.alpha = update

    def update_xi(self, sig_xi, lam_xi=None, sig_xi_terminal=None):
        """The reference copies alpha sig_xi0 into every cell (i2c.py:976-981). Here the cells READ the graph's temperature, so the
        only thing to do is to refuse a noise that is not the current temperature times sig_xi0."""
        if not np.allclose(np.asarray(sig_xi, float), np.asarray(self.sig_xi, float), rtol=1e-12, atol=0.0):
            raise ValueError("update_xi: sig_xi must be alpha * sig_xi0 (set graph.alpha instead)")

    # ------------------------------------------------------------------ observation covariances (i2c.py:983-992)
    def get_z_terminal_covar(self):
        mzt, szt = self.engine.terminal_observed_marginal()
        if mzt is None:
            raise AttributeError("no terminal observation on this graph")
        mzt, szt = np.asarray(mzt.cpu() if hasattr(mzt, "cpu") else mzt, float), np.asarray(szt.cpu() if hasattr(szt, "cpu") else szt, float)
        err = np.asarray(self.z_term, float).reshape(1, -1) - mzt.reshape(self.B, -1)
        return self._per_traj(err[:, :, None] * err[:, None, :] + szt.reshape(self.B, err.shape[1], err.shape[1]))

    def _observed_moments(self, mu, sig):
        """(B, T, d), (B, T, d, d) -> observation moments (B, T, nz), (B, T, nz, nz) through the graph's rule, on the host."""
        B, T = mu.shape[:2]
        nz = self.sys.dim_z
        mz, sz = np.empty((B, T, nz)), np.empty((B, T, nz, nz))
        for b in range(B):
            for t in range(T):
                m, s = self.obs_inf.forward(self.sys.observe, mu[b, t].reshape(-1, 1), sig[b, t])
                mz[b, t], sz[b, t] = np.reshape(m, -1), s
        return mz, sz

    def get_z_propagated_covar(self):
        """sum_t E[(z_t - z*)(z_t - z*)^T] under the closed-loop propagation (i2c.py:986-987, 685-688)."""
        mz, sz = self._table("mu_z0_pf"), self._table("sig_z0_pf")
        err = self._targets() - mz
        return self._per_traj((err[..., :, None] * err[..., None, :] + sz).sum(axis=1))

    # ------------------------------------------------------------------ cost helpers (i2c.py:1029-1070)
    def compute_cost(self, x, u):
        """Quadratic cost of one state-action pair about the graph's target (i2c.py:1029-1032)."""
        xu = np.concatenate((np.reshape(x, -1), np.reshape(u, -1))).reshape(1, -1)
        err = np.reshape(self.sys.observe(xu), (-1, 1)) - np.reshape(self.z, (-1, 1))
        return (err.T @ self.QR @ err).item()

    def compute_cost_gaussian(self, mu_xu, sig_xu):
        """Mean and variance of the quadratic cost of N(mu_xu, sig_xu) pushed through the observation (i2c.py:1034-1043)."""
        mu_z, sig_z = self.obs_inf.forward(self.sys.observe, np.reshape(mu_xu, (-1, 1)), np.asarray(sig_xu, float))
        err = np.reshape(mu_z, (-1, 1)) - np.reshape(self.z, (-1, 1))
        sq = sig_z @ self.QR
        mean = (err.T @ self.QR @ err).item() + float(np.trace(sq))
        var = 2.0 * float(np.trace(sq @ sq)) + 4.0 * (err.T @ (self.QR @ sq) @ err).item()
        return mean, var

    def calc_cost(self):
        """Appends the expected cost (and its variance) of the current posterior -- and of the propagation -- to the cost lists
        (i2c.py:1045-1066), from the statistics the backward / propagation kernels left on the device."""
        self.engine.record_costs()

    @property
    def cost_pf_entropy(self):
        return 0.5 * np.log(_TWO_PI_E * np.asarray(self.costs_pf_var, float))

    @property
    def propagate_cost_improved(self):
        c = self.costs_pf
        if len(c) > 1:
            return bool(np.all(np.asarray(c[-1]) <= np.asarray(c[-2])))
        return True

    # ------------------------------------------------------------------ entropies (i2c.py:1072-1133)
    def _sig_eta_cells(self):
        return np.broadcast_to(np.asarray(self.sys.sig_eta, float), (self.B, self.H) + np.shape(self.sys.sig_eta))

    def calc_policy_entropy(self):
        return self._maybe_scalar(_sum_gaussian_entropy(self._table("sig_u0_m"), "calc_policy_entropy"))

    def calc_sig_eta_entropy(self):
        sig = self._sig_eta_cells()
        self.sig_eta_entropies = list(0.5 * np.linalg.slogdet(_TWO_PI_E * sig[0])[1])
        return self._maybe_scalar(_sum_gaussian_entropy(sig, "calc_sig_eta_entropy"))

    def calc_sig_eta_pf_entropy(self):
        sig = self._sig_eta_cells()  # (known models: the propagation's process noise is the model's, i2c.py:138, 194)
        self.sig_eta_pf_entropies = list(0.5 * np.linalg.slogdet(_TWO_PI_E * sig[0])[1])
        return self._maybe_scalar(_sum_gaussian_entropy(sig, "calc_sig_eta_pf_entropy"))

    def _max_det(self, sig):
        det = np.linalg.det(sig[0])
        t = int(np.argmax(det))
        return float(det[t]), np.array(sig[0][t]), t

    def calc_sig_eta_entropy_max(self):
        return self._max_det(self._sig_eta_cells())

    def calc_sig_eta_pf_entropy_max(self):
        return self._max_det(self._sig_eta_cells())

    def calc_sig_x_pf_entropy_max(self):
        """(largest det of the propagated state covariances, that covariance, its cell). The reference returns the cell's
        `sig_x3_eta` (i2c.py:1098-1101), an attribute no cell has; the covariance itself is returned here."""
        return self._max_det(self._table("sig_x3_pf"))

    def calc_sig_eta_bound_check(self):
        det_pf = np.linalg.det(self._sig_eta_cells()[0])
        det = np.linalg.det(self._sig_eta_cells()[0])
        t = int(np.argmax(det_pf))
        return bool(np.all(1.1 * det[: t + 1] < np.max(det_pf))), np.array(self._sig_eta_cells()[0][t])

    def calc_sig_x_prior_entropy(self):
        return self._maybe_scalar(_sum_gaussian_entropy(self._table("sig_x3_f"), "calc_sig_x_prior_entropy"))

    def calc_propagate_entropy(self):
        return self._maybe_scalar(_sum_gaussian_entropy(self._table("sig_x3_pf"), "calc_propagate_entropy"))

    def _append_iteration_metrics(self):
        """What _maximize appends next to the costs, alpha and the KL term (i2c.py:1021-1027) -- deferred: the packed covariance rows
        the entropies depend on are copied ON THE DEVICE (asynchronously, on the engine's stream); _materialise_metrics() turns the
        pending snapshots into list entries when one of the lists is read."""
        e = self.engine
        d, nx = e.d, e.nx
        o = d + sym_size(d) + nx
        # raw [T][rows][B] slices in the buffers' own row order (a sum over the horizon does not care where the ring starts)
        fwd = e.fwd.permute(0, 2, 1) if e.fwd_trajectory_major else e.fwd
        snap = [e.post[:, d: d + sym_size(d)].clone(),  # (e.post is logically [T][e][B] whatever its storage layout)
                fwd[:, o: o + sym_size(nx)].clone(),
                e.prop[:, o: o + sym_size(nx)].clone() if (self._propagate and e.prop is not None) else None]
        pend = self.__dict__.setdefault("_pending_metrics", [])
        pend.append(snap)
        nbytes = sum(x.numel() * x.element_size() for x in snap if x is not None)
        if len(pend) * nbytes > _MAX_PENDING_BYTES:
            self._materialise_metrics()

    def _drop_last_iteration_metrics(self):
        """Take back what the last _append_iteration_metrics() added: the snapshots are enqueued BEFORE the iteration's one
        synchronisation (they then run while the host waits), and an iteration that raises must leave no entry -- the reference
        appends after its alpha check (i2c.py:1004-1027)."""
        pend = self.__dict__.get("_pending_metrics")
        if pend:
            pend.pop()
            return
        for n in _LAZY_LISTS:  # (the snapshot was materialised on the spot: the pending buffer had reached its cap)
            lst = self.__dict__.get("_m_" + n)
            if lst and (n != "propagate_entropy" or (self._propagate and self.engine.prop is not None)):
                lst.pop()

    def _materialise_metrics(self):
        pending = self.__dict__.get("_pending_metrics")
        if not pending:
            return
        self._pending_metrics = []
        lists = {n: self.__dict__.setdefault("_m_" + n, []) for n in _LAZY_LISTS}
        d, nx = self.engine.d, self.engine.nx
        rows = lambda x, n: unpack_sym(x.permute(2, 0, 1).cpu(), n).numpy()  # noqa: E731  [T][sym][B] -> (B, T, n, n)
        for post_sig, s3f, s3pf in pending:
            sig_u = rows(post_sig, d)[..., nx:, nx:]
            lists["policy_entropy"].append(self._maybe_scalar(_sum_gaussian_entropy(sig_u, "calc_policy_entropy")))
            lists["sig_eta_entropy"].append(self.calc_sig_eta_entropy())
            lists["sig_eta_pf_entropy"].append(self.calc_sig_eta_pf_entropy())
            h = self._maybe_scalar(_sum_gaussian_entropy(rows(s3f, nx), "calc_sig_x_prior_entropy"))
            lists["x_prior_entropy"].append(h)
            lists["x_prior_neg_entropy"].append(-h)
            if s3pf is not None:
                lists["propagate_entropy"].append(self._maybe_scalar(_sum_gaussian_entropy(rows(s3pf, nx), "calc_propagate_entropy")))

    # ------------------------------------------------------------------ likelihood (i2c.py:690-719, 1135-1170)
    def _calc_likelihood(self):
        """(total, state-action, cost, trajectory) log-likelihood terms of the current posterior, as the reference defines them
        (i2c.py:1135-1157 -- determinants where log-determinants would be expected included). Sigma-point graphs push the posterior
        through `sys.forward` on the host with the graph's rule (i2c.py:690-705); single-trajectory graphs only."""
        if self.B != 1:
            raise NotImplementedError("calc_likelihood: single-trajectory graphs (the reference's use)")
        if not isinstance(self.inference, (CubatureQuadrature, GaussHermiteQuadrature)):
            raise NotImplementedError("calc_likelihood under Linearize() needs the cells' linearisations, which stay on the device")
        nx, H = self.sys.dim_x, self.H
        dyn = QuadratureInference(self.inference, self.sys.dim_xu)
        mu, sig = self._table("mu_xu0_m")[0], self._table("sig_xu0_m")[0]
        m3, s3 = self._table("mu_x3_m")[0], self._table("sig_x3_m")[0]
        jx = self._table("Jx_dyn")[0]
        sig_xi = np.asarray(self.sig_xi, float)
        lam_xi = np.linalg.inv(sig_xi)
        mz, sz = self._table("mu_z0_m")[0], self._table("sig_z0_m")[0]
        zt = self._targets()[0]
        m_xu, m_z = np.zeros((nx, nx)), np.zeros((self.sys.dim_z, self.sys.dim_z))
        det_eta = 0.0
        for t in range(H):
            # the action block stands alone in the reference's joint here (concat_normals, i2c.py:691-693)
            s_joint = np.zeros_like(sig[t])
            s_joint[:nx, :nx], s_joint[nx:, nx:] = sig[t][:nx, :nx], sig[t][nx:, nx:]
            mx, sx, eta = dyn.forward_gaussian(self.sys.forward, mu[t].reshape(-1, 1), s_joint)
            mx = np.reshape(mx, -1)
            lag = jx[t] @ s3[t]
            m11 = np.outer(m3[t], m3[t]) + s3[t]
            m01 = np.outer(mx, m3[t]) + lag
            m00 = np.outer(mx, mx) + sx
            m_xu += np.linalg.solve(eta, m00 - m01 - m01.T + m11)
            err = zt[t] - mz[t]
            m_z += lam_xi @ (np.outer(err, err) + sz[t])
            det_eta += np.linalg.det(eta)
        ll_const = -0.5 * H * (nx + self.sys.dim_z) * np.log(2.0 * np.pi)
        ll_sig_xi = -0.5 * H * np.linalg.det(sig_xi)
        ll_sig_eta = -0.5 * det_eta
        ll_sig_x0 = -0.5 * np.linalg.det(np.asarray(self.sys.sig_x0, float))
        ll_xu, ll_z = -0.5 * np.trace(m_xu), -0.5 * np.trace(m_z)
        d0 = mu[0][:nx] - np.asarray(self.sys.x0, float).reshape(-1)
        ll_mu_x0 = -0.5 * np.trace(np.linalg.solve(np.asarray(self.sys.sig_x0, float), np.outer(d0, d0) + sig[0][:nx, :nx]))
        ll_state_action, ll_cost = ll_sig_eta + ll_xu, ll_sig_xi + ll_z
        return ll_const + ll_cost + ll_state_action + ll_sig_x0 + ll_mu_x0, ll_state_action, ll_cost, ll_xu

    def calc_likelihood(self):
        ll, _, ll_z, ll_xu = self._calc_likelihood()  # (the reference's unpacking, i2c.py:1160: the second entry is overwritten)
        self.likelihoods.append(ll)
        self.likelihoods_xu.append(ll_xu)
        self.likelihoods_z.append(ll_z)
        self.risk.append(-2.0 * ll_xu / float(np.asarray(self.alpha)))

    def likelihood_z_minima(self, n_min, n_steps):
        return self.list_minima(self.likelihoods_z, n_min, n_steps)

    def likelihood_xu_minima(self, n_min, n_steps):
        return self.list_minima(self.likelihoods_xu, n_min, n_steps)

    # ------------------------------------------------------------------ misc (i2c.py:1223-1229, 1294-1298)
    @staticmethod
    def mvn_kl_divergence(mu1, sig1, mu2, sig2):
        """KL(N(mu1, sig1) || N(mu2, sig2))."""
        mu1, mu2 = np.reshape(mu1, -1).astype(float), np.reshape(mu2, -1).astype(float)
        sig1, sig2 = np.asarray(sig1, float), np.asarray(sig2, float)
        d = mu2 - mu1
        l1, l2 = np.linalg.slogdet(sig1)[1], np.linalg.slogdet(sig2)[1]
        return 0.5 * ((l2 - l1) + np.trace(np.linalg.solve(sig2, sig1)) + float(d @ np.linalg.solve(sig2, d)) - mu1.shape[0])

    def get_prior_state_action_distribution(self):
        """The joint prior the LAST forward sweep started from -- what the reference keeps as mu_xu0_f_prev / sig_xu0_f_prev when
        _update_priors overwrites the prior with the posterior (i2c.py:1216-1219, 1294-1298)."""
        return self._squeeze(self._table("mu_xu0_f")), self._squeeze(self._table("sig_xu0_f"))
