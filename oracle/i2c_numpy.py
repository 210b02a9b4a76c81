"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy, fp64) restatement of the reference's Gaussian i2c
cubature hot path, batched over a leading trajectory axis B.

  * Never imported by the product path (input-inference-for-control_amd/). Only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it -- as the checker
    or the timed CPU baseline, never as the thing shipped.
  * Parity status: PINNED. tests/test_oracle_golden.py checks every function below against
    vectors captured from the real reference imported in-container
    (oracle/gen_golden.py -> tests/golden/*.npz).
  * Written from the behaviour spec in SURVEY.md Appendix A; it follows the reference's
    *formulas* (e.g. the uncentred covariance  sum w y y^T - m m^T) so that it agrees with
    the reference to rounding, while the HIP kernels use numerically kinder but
    mathematically identical forms.

Citations are relative to /root/reference. Conventions: means are (B, n) row arrays (the
reference uses (n, 1) columns), covariances (B, n, n); per-cell state is (B, T, ...).
"""
import numpy as np


# --------------------------------------------------------------------------------------
# quadrature rules (i2c/exp_types.py:30-68)
# --------------------------------------------------------------------------------------
class CubatureRule:
    """Scaled symmetric sigma-point rule (exp_types.py:36-49): P = 2 dim + 1 points."""

    def __init__(self, alpha=1.0, beta=0.0, kappa=0.0):
        assert alpha > 0
        self.alpha, self.beta, self.kappa = float(alpha), float(beta), float(kappa)

    def points(self, dim):
        eye = np.eye(dim)
        return np.concatenate((np.zeros((1, dim)), eye, -eye), axis=0)

    def weights(self, dim):
        lam = self.alpha ** 2 * (dim + self.kappa) - dim
        sf = np.sqrt(dim + lam)
        w_mu = np.full(2 * dim + 1, 1.0 / (2.0 * (dim + lam)))
        w_mu[0] = 2.0 * lam * w_mu[0]
        w_sig = w_mu.copy()
        w_sig[0] += 1.0 - self.alpha ** 2 + self.beta
        return sf, w_mu, w_sig


class GaussHermiteRule:
    """Tensor-grid Gauss-Hermite rule (exp_types.py:52-68): P = degree ** dim points."""

    def __init__(self, degree):
        assert degree >= 1
        self.degree = int(degree)
        self._x, self._w = np.polynomial.hermite.hermgauss(self.degree)

    def points(self, dim):
        grid = np.meshgrid(*(self._x,) * dim)
        return np.vstack([g.ravel() for g in grid]).T

    def weights(self, dim):
        grid = np.meshgrid(*(self._w,) * dim)
        w = np.prod(np.vstack([g.ravel() for g in grid]).T, axis=1) / np.pi ** (dim / 2)
        return np.sqrt(2.0), w, w


class SigmaPointTransform:
    """Gaussian -> Gaussian push-through (i2c/inference/quadrature.py:7-58), batched."""

    def __init__(self, rule, dim):
        self.dim = dim
        self.base = rule.points(dim)  # (P, dim)
        self.sf, self.w_mu, self.w = rule.weights(dim)  # all moments use w_sig (quadrature.py:36,49)

    def points(self, m, S):
        """quadrature.py:15-25 : X = m + base @ (sf L)^T with L = chol_lower(S)."""
        L = np.linalg.cholesky(S)  # raises LinAlgError if any batch member is not PD
        return m[..., None, :] + np.einsum("pj,...ij->...pi", self.base, self.sf * L)

    def moments(self, m, X, Y):
        """quadrature.py:34-44 : weighted mean, covariance, cross-covariance (uncentred form)."""
        w = self.w
        m_y = np.einsum("p,...pi->...i", w, Y)
        S_y = np.einsum("p,...pi,...pj->...ij", w, Y, Y) - m_y[..., :, None] * m_y[..., None, :]
        S_xy = np.einsum("p,...pi,...pj->...ij", w, X, Y) - m[..., :, None] * m_y[..., None, :]
        return m_y, S_y, S_xy

    def forward(self, f, m, S):
        """quadrature.py:27-32. Returns (m_y, S_y, S_xy, X, Y)."""
        X = self.points(m, S)
        Y = f(X)
        m_y, S_y, S_xy = self.moments(m, X, Y)
        return m_y, S_y, S_xy, X, Y


def _sym_solve(A, Bm):
    """X = Bm A^{-1} for SPD A, as the reference does  la.solve(A.T, Bm.T, assume_a='pos').T."""
    return np.swapaxes(np.linalg.solve(np.swapaxes(A, -1, -2), np.swapaxes(Bm, -1, -2)), -1, -2)


def _outer(a, b):
    return a[..., :, None] * b[..., None, :]


def _mv(M, v):
    return np.einsum("...ij,...j->...i", M, v)


def _T(M):
    return np.swapaxes(M, -1, -2)


class I2cOracle:
    """Batched restatement of I2cGraph + I2cCell (cubature path), i2c/i2c.py:51-148, 350-447,
    544-610, 732-1065, 1210-1251.

    Extra (build-defined, the reference has B = 1): leading batch axis; per-trajectory
    x0 (B, nx), mu_u (B, T, nu), alpha (B,), and optional per-cell targets z (B, T, nz).
    """

    def __init__(self, model, horizon, Q, R, Qf, alpha, alpha_update_tol, mu_u, sig_u,
                 mu_x_terminal=None, sig_x_terminal=None, rule=None, x0=None, sig_x0=None,
                 z_traj=None, dtemp=1.0):
        self.sys = model
        self.H = T = int(horizon)
        nx, nu, nz = model.dim_x, model.dim_u, model.dim_z
        self.nx, self.nu, self.nz, self.d = nx, nu, nz, nx + nu
        d = self.d
        mu_u = np.asarray(mu_u, dtype=float)
        if mu_u.ndim == 2:
            mu_u = mu_u[None]
        B = mu_u.shape[0]
        if x0 is not None:
            x0 = np.asarray(x0, dtype=float).reshape(-1, nx)
            B = max(B, x0.shape[0])
        self.B = B
        mu_u = np.broadcast_to(mu_u, (B, T, nu)).copy()
        self.x0 = np.broadcast_to(model.x0 if x0 is None else x0, (B, nx)).copy()
        self.sig_x0 = np.broadcast_to(model.sig_x0 if sig_x0 is None else sig_x0, (B, nx, nx)).copy()
        self.sig_eta = np.asarray(model.sig_eta, dtype=float)
        self.rule = rule if rule is not None else CubatureRule(1, 0, 0)

        # cost "observation" model (i2c.py:778-793)
        R = np.atleast_2d(np.asarray(R, dtype=float))
        if Q is not None:
            Q = np.atleast_2d(np.asarray(Q, dtype=float))
            self.QR = np.block([[Q, np.zeros((Q.shape[0], nu))], [np.zeros((nu, Q.shape[0])), R]])
        else:
            self.QR = R
        assert self.QR.shape == (nz, nz)
        self.sig_xi0 = np.linalg.inv(self.QR)
        if Qf is not None:
            self.Qf = np.atleast_2d(np.asarray(Qf, dtype=float))
            self.sig_xi_terminal_base = np.linalg.inv(self.Qf)
        else:
            self.Qf = None
            self.sig_xi_terminal_base = None
        self.alpha = np.broadcast_to(np.asarray(alpha, dtype=float), (B,)).copy()
        self.alpha_update_tol = float(alpha_update_tol)
        self.alphas = [self.alpha.copy()]
        self.alphas_desired = [self.alpha.copy()]
        self.alphas_pf = [self.alpha.copy()]

        # targets (i2c.py:84-85; MPC overrides per cell, mpc.py:29-31)
        if z_traj is None:
            self.z = np.broadcast_to(np.asarray(model.zg, dtype=float).reshape(-1), (B, T, nz)).copy()
        else:
            self.z = np.broadcast_to(np.asarray(z_traj, dtype=float), (B, T, nz)).copy()
        self.z_term = None if model.zg_term is None else np.asarray(model.zg_term, dtype=float).reshape(-1)

        # covariance control (i2c.py:797-802, 145-148)
        self.mu_x_terminal = None if mu_x_terminal is None else np.asarray(mu_x_terminal, dtype=float).reshape(nx)
        self.sig_x_terminal = None if sig_x_terminal is None else np.asarray(sig_x_terminal, dtype=float)
        self.temp = np.ones(B)
        self.dtemp = float(dtemp)

        # transforms (i2c.py:116-121, 839-840)
        self.tf_xu = SigmaPointTransform(self.rule, d)
        self.tf_x = SigmaPointTransform(self.rule, nx)

        # per-cell state (I2cCell.__init__, i2c.py:81-148)
        sig_u = np.atleast_2d(np.asarray(sig_u, dtype=float))
        self.mu_u0_f = mu_u.copy()
        self.sig_u0_f = np.broadcast_to(sig_u, (B, T, nu, nu)).copy()
        self.mu_xu0_f = np.concatenate((np.broadcast_to(self.x0[:, None, :], (B, T, nx)), mu_u), axis=-1)
        blk = np.zeros((B, T, d, d))
        blk[:, :, :nx, :nx] = self.sig_x0[:, None]
        blk[:, :, nx:, nx:] = sig_u
        self.sig_xu0_f = blk
        self.mu_xu0_m = self.mu_xu0_f.copy()
        self.sig_xu0_m = self.sig_xu0_f.copy()
        self.K = np.zeros((B, T, nu, nx))
        self.k = mu_u.copy()
        self.sigK = np.broadcast_to(sig_u, (B, T, nu, nu)).copy()
        self.feedforward = np.ones(T, dtype=bool)  # state_action_independence (i2c.py:132)
        self.use_expert_controller = True
        self.tau = T - 1  # i2c.py:833
        self._propagate = False

        z = lambda *s: np.zeros((B, T) + s)
        self.mu_xu1_f, self.sig_xu1_f = z(d), z(d, d)
        self.mu_x3_f, self.sig_x3_f, self.J_dyn = z(nx), z(nx, nx), z(d, nx)
        self.mu_z0_f, self.sig_z0_f = z(nz), z(nz, nz)
        self.mu_z0_m, self.sig_z0_m = z(nz), z(nz, nz)
        self.mu_x3_m, self.sig_x3_m = z(nx), z(nx, nx)
        self.mu_z3_m = self.sig_z3_m = None
        # closed-loop propagation state (i2c.py:138-141)
        self.mu_xu0_pf, self.sig_xu0_pf = z(d), z(d, d)
        self.mu_z0_pf, self.sig_z0_pf = z(nz), z(nz, nz)
        self.mu_x3_pf = np.broadcast_to(self.x0[:, None, :], (B, T, nx)).copy()
        self.sig_x3_pf = np.broadcast_to(self.sig_x0[:, None], (B, T, nx, nx)).copy()

        self.costs_m, self.costs_m_var, self.costs_pf, self.costs_pf_var = [], [], [], []
        self.kl_terms = []
        self.em_iter = 0

    # ------------------------------------------------------------------ temperatures
    @property
    def sig_xi(self):  # (B, nz, nz)   i2c.py:853-855
        return self.alpha[:, None, None] * self.sig_xi0

    @property
    def sig_xi_terminal(self):  # i2c.py:857-862
        if self.sig_xi_terminal_base is None:
            return None
        return self.alpha[:, None, None] * self.sig_xi_terminal_base

    def _forward_model(self, xu):
        return self.sys.dynamics(xu)

    # ------------------------------------------------------------------ forward cell
    def _forward_cell(self, t, mu_x, sig_x):
        """I2cCell._forward_msgs_quadrature, i2c.py:350-447."""
        nx = self.nx
        if self.feedforward[t]:  # i2c.py:355-360
            mu0 = np.concatenate((mu_x, self.mu_u0_f[:, t]), axis=-1)
            S0 = np.zeros((self.B, self.d, self.d))
            S0[:, :nx, :nx] = sig_x
            S0[:, nx:, nx:] = self.sig_u0_f[:, t]
        else:  # i2c.py:361-387
            pj_mu, pj_sig = self.mu_xu0_f[:, t], self.sig_xu0_f[:, t]
            sig_xx, sig_ux = pj_sig[:, :nx, :nx], pj_sig[:, nx:, :nx]
            S = sig_xx + sig_x
            delta = mu_x - pj_mu[:, :nx]
            # N(mu_x; m, S) / N(m; m, S): normalisers cancel (i2c.py:369-374)
            maha = np.einsum("bi,bi->b", delta, np.linalg.solve(S, delta[..., None])[..., 0])
            np.linalg.cholesky(S)  # scipy mvn(allow_singular=False) raises on singular S
            K = self.K[:, t] * np.exp(-0.5 * maha)[:, None, None]
            mu_x0_m, mu_u0_m = self.mu_xu0_m[:, t, :nx], self.mu_xu0_m[:, t, nx:]
            sig_u0_m = self.sig_xu0_m[:, t, nx:, nx:]
            mu_u = mu_u0_m + _mv(K, mu_x - mu_x0_m)
            sig_u = sig_u0_m - K @ _T(sig_ux) + K @ sig_x @ _T(K)
            self.mu_u0_f[:, t], self.sig_u0_f[:, t] = mu_u, sig_u
            mu0 = np.concatenate((mu_x, mu_u), axis=-1)
            S0 = np.concatenate(
                (np.concatenate((sig_x, sig_x @ _T(K)), axis=-1), np.concatenate((K @ sig_x, sig_u), axis=-1)),
                axis=-2,
            )
        self.mu_xu0_f[:, t], self.sig_xu0_f[:, t] = mu0, S0

        # cost observation = Kalman-style measurement update (i2c.py:390-407)
        mu_z, sig_z, sig_xz, _, _ = self.tf_xu.forward(self.sys.observe, mu0, S0)
        sig_z = sig_z + self.sig_xi
        G = _sym_solve(sig_z, sig_xz)
        mu1 = mu0 + _mv(G, self.z[:, t] - mu_z)
        S1 = S0 - G @ _T(sig_xz)
        self.mu_z0_f[:, t], self.sig_z0_f[:, t] = mu_z, sig_z
        self.mu_xu1_f[:, t], self.sig_xu1_f[:, t] = mu1, S1

        # dynamics (i2c.py:415-421); known models: sig_noise = sum_p w_p sig_eta (quadrature.py:57)
        mu3, sig_y, sig_xy, _, _ = self.tf_xu.forward(self._forward_model, mu1, S1)
        sig_noise = self.tf_xu.w.sum() * self.sig_eta
        sig3 = sig_y + sig_noise
        sig3 = (sig3 + _T(sig3)) / 2
        # smoother gain (i2c.py:423-425)
        J = _sym_solve(sig3, sig_xy)

        # terminal cost observation on the LAST cell only, after J (i2c.py:430-443)
        if t == self.H - 1 and self.sig_xi_terminal is not None:
            mu_zt, sig_zt, sig_xzt, _, _ = self.tf_x.forward(self.sys.observe_terminal, mu3, sig3)
            sig_zt = sig_zt + self.sig_xi_terminal
            Gt = _sym_solve(sig_zt, sig_xzt)
            mu3 = mu3 + _mv(Gt, self.z_term - mu_zt)
            sig3 = sig3 - Gt @ _T(sig_xzt)
        self.mu_x3_f[:, t], self.sig_x3_f[:, t], self.J_dyn[:, t] = mu3, sig3, J
        return mu3, sig3

    def forward_sweep(self):
        """I2cGraph._forward_msgs, i2c.py:876-880."""
        mu, sig = self.x0.copy(), self.sig_x0.copy()
        for t in range(self.H):
            mu, sig = self._forward_cell(t, mu, sig)

    # ------------------------------------------------------------------ backward cell
    def _backward_cell(self, t, mu_end, sig_end):
        """I2cCell._backward_msgs_quadrature, i2c.py:544-610."""
        nx = self.nx
        if mu_end is None:  # end of chain (i2c.py:546-572)
            mu3f, sig3f = self.mu_x3_f[:, t], self.sig_x3_f[:, t]
            if self.sig_x_terminal is not None:  # covariance control, tempered prior (i2c.py:548-559)
                sig_t = self.temp[:, None, None] * sig3f
                self.temp = self.temp + self.dtemp
                sig3m = sig_t - sig_t @ np.linalg.solve(self.sig_x_terminal + sig_t, sig_t)
                rhs = np.linalg.solve(sig_t, mu3f[..., None])[..., 0] + np.linalg.solve(
                    self.sig_x_terminal, self.mu_x_terminal
                )
                mu3m = _mv(sig3m, rhs)
            else:
                mu3m, sig3m = mu3f, sig3f
            if self.sig_xi_terminal is not None:  # i2c.py:567-570 (no +sig_xi_terminal here)
                self.mu_z3_m, self.sig_z3_m, _, _, _ = self.tf_x.forward(self.sys.observe_terminal, mu3m, sig3m)
            else:
                self.mu_z3_m = self.sig_z3_m = None
        else:
            mu3m, sig3m = mu_end, sig_end
        self.mu_x3_m[:, t], self.sig_x3_m[:, t] = mu3m, sig3m

        J = self.J_dyn[:, t]
        mu_m = self.mu_xu1_f[:, t] + _mv(J, mu3m - self.mu_x3_f[:, t])  # i2c.py:580
        sig_m = self.sig_xu1_f[:, t] + J @ (sig3m - self.sig_x3_f[:, t]) @ _T(J)  # i2c.py:581-583
        self.mu_xu0_m[:, t], self.sig_xu0_m[:, t] = mu_m, sig_m

        # posterior observation statistics for the M-step (i2c.py:594-596)
        self.mu_z0_m[:, t], self.sig_z0_m[:, t], _, _, _ = self.tf_xu.forward(self.sys.observe, mu_m, sig_m)

        # time-varying linear-Gaussian controller p(u | x) (i2c.py:600-608)
        sig_xx, sig_ux = sig_m[:, :nx, :nx], sig_m[:, nx:, :nx]
        K = _sym_solve(sig_xx, sig_ux)
        self.K[:, t] = K
        self.k[:, t] = mu_m[:, nx:] - _mv(K, mu_m[:, :nx])
        self.sigK[:, t] = sig_m[:, nx:, nx:] - K @ _T(sig_ux)
        return mu_m[:, :nx], sig_xx

    def backward_sweep(self):
        """I2cGraph._backward_msgs, i2c.py:882-886."""
        mu = sig = None
        for t in reversed(range(self.H)):
            mu, sig = self._backward_cell(t, mu, sig)

    def forward_backward(self):
        self.forward_sweep()
        self.backward_sweep()

    # ------------------------------------------------------------------ closed-loop propagation
    def _pdf_ratio(self, mean, cov, x):
        delta = x - mean
        np.linalg.cholesky(cov)
        return np.exp(-0.5 * np.einsum("bi,bi->b", delta, np.linalg.solve(cov, delta[..., None])[..., 0]))

    def _propagate_cell(self, t, mu_x, sig_x):
        """I2cCell._propagate_forward_quadrature, i2c.py:150-199."""
        nx = self.nx
        K = self.K[:, t].copy()
        mu_x0_m, mu_u0_m = self.mu_xu0_m[:, t, :nx], self.mu_xu0_m[:, t, nx:]
        sig_x0_m, sig_u0_m = self.sig_xu0_m[:, t, :nx, :nx], self.sig_xu0_m[:, t, nx:, nx:]
        if self.feedforward[t]:  # i2c.py:155-157
            mu_u, sig_u = mu_u0_m, sig_u0_m
        else:  # i2c.py:158-171
            if self.use_expert_controller:
                K = K * self._pdf_ratio(mu_x0_m, sig_x0_m + sig_x, mu_x)[:, None, None]
            mu_u = mu_u0_m + _mv(K, mu_x - mu_x0_m)
            sig_u = K @ sig_x @ _T(K) + sig_u0_m - K @ sig_x0_m @ _T(K)
        mu0 = np.concatenate((mu_x, mu_u), axis=-1)
        S0 = np.concatenate(
            (np.concatenate((sig_x, sig_x @ _T(K)), axis=-1), np.concatenate((K @ sig_x, sig_u), axis=-1)), axis=-2
        )
        self.mu_xu0_pf[:, t], self.sig_xu0_pf[:, t] = mu0, S0
        self.mu_z0_pf[:, t], self.sig_z0_pf[:, t], _, _, _ = self.tf_xu.forward(self.sys.observe, mu0, S0)
        mu3, sig_y, _, _, _ = self.tf_xu.forward(self._forward_model, mu0, S0)
        sig3 = sig_y + self.tf_xu.w.sum() * self.sig_eta  # i2c.py:195 (not symmetrised here)
        self.mu_x3_pf[:, t], self.sig_x3_pf[:, t] = mu3, sig3
        return mu3, sig3

    def propagate(self):
        """I2cGraph.propagate, i2c.py:1247-1251."""
        mu, sig = self.x0.copy(), self.sig_x0.copy()
        for t in range(self.H):
            mu, sig = self._propagate_cell(t, mu, sig)

    # ------------------------------------------------------------------ M-step
    def _z_covar(self, mu_z, sig_z):
        """sum_t [(z - mu_z)(z - mu_z)^T + sig_z]  (i2c.py:680-688, 983-987)."""
        err = self.z - mu_z
        return (_outer(err, err) + sig_z).sum(axis=1)

    def calculate_alpha(self, z_covar, z_covar_term=None):
        """i2c.py:913-919."""
        tr = np.einsum("ij,bji->b", self.QR, z_covar)
        sf = float(self.nz * self.H)
        if z_covar_term is not None:
            tr = tr + np.einsum("ij,bji->b", self.Qf, z_covar_term)
            sf += float(self.sys.dim_z_term)
        return tr / sf

    def compute_update_alpha(self, update_alpha=True):
        """i2c.py:921-963."""
        z_covar = self._z_covar(self.mu_z0_m, self.sig_z0_m)
        z_covar_term = None
        if self.sig_xi_terminal_base is not None:  # i2c.py:989-992
            err = self.z_term - self.mu_z3_m
            z_covar_term = _outer(err, err) + self.sig_z3_m
        alpha_hat = self.calculate_alpha(z_covar, z_covar_term)
        if self._propagate:
            self.alphas_pf.append(self.calculate_alpha(self._z_covar(self.mu_z0_pf, self.sig_z0_pf)))
        self.alphas_desired.append(alpha_hat.copy())
        if update_alpha:
            if np.any(np.isnan(alpha_hat)):
                raise ValueError("Alpha is NaN")
            tol = self.alpha_update_tol
            if tol >= 0.0:
                ratio = alpha_hat / self.alpha
                new = np.where(ratio < tol, tol * self.alpha, alpha_hat)
                new = np.where(ratio > 2.0 - tol, (2.0 - tol) * self.alpha, new)
            else:
                new = self.alpha.copy()
            self.alpha = new
        self.alphas.append(self.alpha.copy())

    def calibrate_alpha(self, only_decrease=False):
        """i2c.py:895-911."""
        assert self._propagate
        self.propagate()
        a = self.calculate_alpha(self._z_covar(self.mu_z0_pf, self.sig_z0_pf))
        upd = (a < self.alpha) if only_decrease else np.ones(self.B, dtype=bool)
        self.alpha = np.where(upd, a, self.alpha)
        self.alphas[-1] = self.alpha.copy()

    def _gaussian_cost(self, mu_z, sig_z):
        """I2cGraph.compute_cost_gaussian summed over t (i2c.py:1034-1053)."""
        err = mu_z - self.z
        sq = sig_z @ self.QR
        m = np.einsum("bti,ij,btj->bt", err, self.QR, err) + np.trace(sq, axis1=-2, axis2=-1)
        v = 2 * np.trace(sq @ sq, axis1=-2, axis2=-1) + 4 * np.einsum(
            "bti,btij,btj->bt", err, self.QR @ sq, err
        )
        return m, v

    def calc_cost(self):
        """i2c.py:1045-1065. The reference re-runs the observe transform at (mu_xu0_m, sig_xu0_m);
        that is exactly (mu_z0_m, sig_z0_m) from the backward pass (i2c.py:586-587, 594-596)."""
        m, v = self._gaussian_cost(self.mu_z0_m, self.sig_z0_m)
        self.costs_m.append(m.sum(axis=1))
        self.costs_m_var.append(v.sum(axis=1))
        if self._propagate:
            m, v = self._gaussian_cost(self.mu_z0_pf, self.sig_z0_pf)
            self.costs_pf.append(m.sum(axis=1))
            self.costs_pf_var.append(v.sum(axis=1))
        else:
            self.costs_pf.append(-np.ones(self.B))

    def update_priors(self):
        """I2cGraph._update_priors, i2c.py:1210-1221."""
        if self.tau > 0:
            self.feedforward[: self.tau + 1] = False
        nx = self.nx
        self.mu_u0_f = self.mu_xu0_m[:, :, nx:].copy()
        self.sig_u0_f = self.sig_xu0_m[:, :, nx:, nx:].copy()
        self.mu_xu0_f = self.mu_xu0_m.copy()
        self.sig_xu0_f = self.sig_xu0_m.copy()

    @staticmethod
    def mvn_kl(mu1, sig1, mu2, sig2):
        """i2c.py:1223-1229, batched."""
        diff = mu2 - mu1
        dist = np.einsum("bi,bi->b", diff, np.linalg.solve(sig2, diff[..., None])[..., 0])
        log_det_ratio = np.log(np.linalg.det(sig2) / np.linalg.det(sig1))
        trace_ratio = np.trace(np.linalg.solve(sig2, sig1), axis1=-2, axis2=-1)
        return 0.5 * (log_det_ratio + trace_ratio + dist - mu1.shape[-1])

    def maximize(self):
        """I2cGraph._maximize, i2c.py:1004-1019 (entropy logging omitted)."""
        self.calc_cost()
        self.update_priors()
        self.compute_update_alpha(True)
        if self.sig_x_terminal is not None and self.mu_x_terminal is not None:
            B = self.B
            self.kl_terms.append(
                self.mvn_kl(
                    self.mu_x3_pf[:, -1],
                    self.sig_x3_pf[:, -1],
                    np.broadcast_to(self.mu_x_terminal, (B, self.nx)),
                    np.broadcast_to(self.sig_x_terminal, (B, self.nx, self.nx)),
                )
            )

    def learn_msgs(self):
        """I2cGraph.learn_msgs, i2c.py:1238-1245."""
        self.em_iter += 1
        self.forward_backward()
        if self._propagate:
            self.propagate()
        self.maximize()

    # ------------------------------------------------------------------ getters (i2c.py:1253-1314)
    def get_local_linear_policy(self):
        return self.K.copy(), self.k.copy(), self.sigK.copy()
