"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (NumPy fp64, batched over a leading axis) of the reference's *Linearize* inference path:
I2cCell._forward_msgs_linearize (i2c/i2c.py:244-348), _backward_msgs_linearize (:449-542),
_backward_ricatti_msgs (:612-678) and the pieces of I2cGraph that differ from the cubature path
(cost through the graph's cubature transform, :839-846, 1034-1053; alpha statistics from the linearised
posterior observation, :538-540, 680-683, 989-992). Everything else (priors, M-step, propagation, which
stays a cubature transform in this mode, :109-115) is inherited from oracle/i2c_numpy.py.

Parity status: PINNED for the solver algebra against the real reference on the linear systems (no Jacobian
needed there: LinearBase.forward_linearize, i2c/model.py:227-229) -- tests/golden/lin_*.npz.
For the nonlinear known models the reference takes dynamics Jacobians from `autograd.jacobian`
(i2c/env_autograd.py:22,57,170), which is not installed in this image; the golden vectors of those cases were made with
a complex-step stand-in for it (oracle/ref_shim.py) applied to the reference's own dynamics code, so they pin the
solver against the reference up to the Jacobian's rounding (~1e-15 relative), and say so in their metadata.
"""
import numpy as np

from oracle.i2c_numpy import I2cOracle, _T, _mv, _outer, _sym_solve


def complex_step_jacobian(f, x):
    """d f / d x at real x (..., n) for analytic NumPy code f: (..., n) -> (..., m). Returns (..., m, n)."""
    x = np.asarray(x, dtype=float)
    cols = []
    for j in range(x.shape[-1]):
        xc = x.astype(complex)
        xc[..., j] += 1e-30j
        with np.errstate(all="ignore"):
            cols.append(np.imag(f(xc)) / 1e-30)
    return np.stack(cols, axis=-1)


class I2cLinearizeOracle(I2cOracle):
    """I2cGraph(inference=Linearize()) restated; same constructor as I2cOracle (the `rule` is the graph's
    CubatureQuadrature(1, 0, 0) used for cost evaluation and propagation, i2c.py:841-844, 109-115)."""

    def __init__(self, *a, **k):
        k.pop("rule", None)
        super().__init__(*a, **k)
        B, T, nx, nu, nz, d = self.B, self.H, self.nx, self.nu, self.nz, self.d
        z = lambda *s: np.zeros((B, T) + s)  # noqa: E731
        self.A, self.Bm, self.a = z(nx, nx), z(nx, nu), z(nx)
        self.E, self.F, self.e = z(nz, nx), z(nz, nu), z(nz)
        # Riccati bookkeeping of the forward pass (i2c.py:278-294, 312-335, 346)
        self.lambda_z1_f, self.nu_z1_f = z(nz, nz), z(nx)
        self.lambda_z2_f, self.nu_z2_f = z(nz, nz), z(nu)
        self.sig_u2_f, self.sig_x2_f, self.lambda_x2_f = z(nx, nx), z(nx, nx), z(nx, nx)
        self.nu_x3_f, self.lambda_x3_f = z(nx), z(nx, nx)
        self.mu_x0_f, self.sig_x0_f = z(nx), z(nx, nx)
        self.nu_x0_b, self.lambda_x0_b = z(nx), z(nx, nx)
        self.mu_z_cost, self.sig_z_cost = z(nz), z(nz, nz)

    # ---- model linearisations ------------------------------------------------------------
    def _observe_lin(self, xu):
        """sys.observe_linearize (i2c/env_def.py:87-91, 278-287, 334-340, 551-570, 697-718): z, E = dz/dx, F = dz/du."""
        z = self.sys.observe(xu)
        J = complex_step_jacobian(self.sys.observe, xu)
        return z, J[..., : self.nx], J[..., self.nx:]

    def _observe_terminal_lin(self, x):
        z = self.sys.observe_terminal(x)
        if z is None:
            raise TypeError("observe_terminal_linearize returns None for this model (i2c/env_def.py:345-346): the "
                            "reference's Linearize path fails on it at i2c.py:500-501")
        return z, complex_step_jacobian(self.sys.observe_terminal, x)

    def _forward_lin(self, xu):
        """BaseModelKnown.forward_linearize (i2c/model.py:158-164): x' = f(xu), AB = df/dxu, a = x' - AB xu."""
        x3 = self.sys.dynamics(xu)
        AB = complex_step_jacobian(self.sys.dynamics, xu)
        return x3, AB[..., : self.nx], AB[..., self.nx:], x3 - _mv(AB, xu)

    # ---- forward cell ----------------------------------------------------------------------
    def _forward_cell(self, t, mu_x, sig_x):
        """I2cCell._forward_msgs_linearize, i2c.py:244-348."""
        nx, B = self.nx, self.B
        self.mu_x0_f[:, t], self.sig_x0_f[:, t] = mu_x, sig_x
        if self.feedforward[t]:  # i2c.py:249-252
            mu_u, sig_u = self.mu_u0_f[:, t], self.sig_u0_f[:, t]
            mu0 = np.concatenate((mu_x, mu_u), axis=-1)
            S0 = np.zeros((B, self.d, self.d))
            S0[:, :nx, :nx] = sig_x
            S0[:, nx:, nx:] = sig_u
        else:  # i2c.py:253-276
            pj_mu, pj_sig = self.mu_xu0_f[:, t], self.sig_xu0_f[:, t]
            sig_xx, sig_ux = pj_sig[:, :nx, :nx], pj_sig[:, nx:, :nx]
            K = self.K[:, t].copy()
            if self.use_expert_controller:  # only then is the gain scaled in this mode (i2c.py:259-265)
                K = K * self._pdf_ratio(pj_mu[:, :nx], sig_xx + sig_x, mu_x)[:, None, None]
            mu_x0_m, mu_u0_m = self.mu_xu0_m[:, t, :nx], self.mu_xu0_m[:, t, nx:]
            sig_u0_m = self.sig_xu0_m[:, t, nx:, nx:]
            mu_u = mu_u0_m + _mv(K, mu_x - mu_x0_m)
            sig_u = sig_u0_m - K @ _T(sig_ux) + K @ sig_x @ _T(K)
            mu0 = np.concatenate((mu_x, mu_u), axis=-1)
            S0 = np.concatenate(
                (np.concatenate((sig_x, sig_x @ _T(K)), axis=-1), np.concatenate((K @ sig_x, sig_u), axis=-1)), axis=-2
            )
        self.mu_u0_f[:, t], self.sig_u0_f[:, t] = mu_u, sig_u
        self.mu_xu0_f[:, t], self.sig_xu0_f[:, t] = mu0, S0

        # observation linearised about the prior mean (i2c.py:281-283)
        mu_z, E, F = self._observe_lin(mu0)
        e = mu_z - _mv(E, mu_x) - _mv(F, mu_u)
        self.E[:, t], self.F[:, t], self.e[:, t] = E, F, e
        sig_xi = self.sig_xi
        # Riccati bookkeeping (i2c.py:278, 285-294)
        lam_x0 = np.linalg.inv(sig_x)
        nu_x0 = _mv(lam_x0, mu_x)
        sig_z1 = sig_xi + F @ sig_u @ _T(F)
        lam_z1 = np.linalg.inv(sig_z1)
        self.lambda_z1_f[:, t] = lam_z1
        self.nu_z1_f[:, t] = _mv(_T(E), _mv(lam_z1, self.z[:, t] - _mv(F, mu_u) - e))
        del nu_x0, lam_x0

        EF = np.concatenate((E, F), axis=-1)
        sig_z = EF @ S0 @ _T(EF) + sig_xi  # i2c.py:296-297
        sig_zxu = EF @ S0
        G = _T(np.linalg.solve(_T(sig_z), sig_zxu))  # i2c.py:299
        mu1 = mu0 + _mv(G, self.z[:, t] - mu_z)
        S1 = S0 - G @ sig_zxu
        self.mu_z0_f[:, t], self.sig_z0_f[:, t] = mu_z, sig_z
        self.mu_xu1_f[:, t], self.sig_xu1_f[:, t] = mu1, S1

        sig_z2 = sig_xi + E @ sig_x @ _T(E)  # i2c.py:312-317
        lam_z2 = np.linalg.inv(sig_z2)
        self.lambda_z2_f[:, t] = lam_z2
        self.nu_z2_f[:, t] = _mv(_T(F), _mv(lam_z2, (self.z[:, t] - _mv(E, mu_x)) - e))

        # dynamics linearised about the updated mean (i2c.py:321-328)
        mu3, A, Bm, a = self._forward_lin(mu1)
        self.A[:, t], self.Bm[:, t], self.a[:, t] = A, Bm, a
        AB = np.concatenate((A, Bm), axis=-1)
        sig3 = AB @ S1 @ _T(AB) + self.sig_eta
        self.sig_u2_f[:, t] = Bm @ S1[:, nx:, nx:] @ _T(Bm)  # i2c.py:330-332
        self.sig_x2_f[:, t] = A @ S1[:, :nx, :nx] @ _T(A) + self.sig_eta
        self.lambda_x2_f[:, t] = np.linalg.inv(self.sig_x2_f[:, t])
        J = _sym_solve(sig3, _T(AB @ S1))  # i2c.py:335-341: la.solve(sig_x3_f.T, AB sig_xu1_f, assume_a="pos").T
        self.lambda_x3_f[:, t] = np.linalg.inv(sig3)  # i2c.py:346
        self.nu_x3_f[:, t] = _mv(self.lambda_x3_f[:, t], mu3)
        self.mu_x3_f[:, t], self.sig_x3_f[:, t], self.J_dyn[:, t] = mu3, sig3, J
        return mu3, sig3

    # ---- backward cell ---------------------------------------------------------------------
    def _backward_cell(self, t, mu_end, sig_end):
        """I2cCell._backward_msgs_linearize, i2c.py:449-542."""
        nx = self.nx
        if mu_end is None:
            mu3f, sig3f = self.mu_x3_f[:, t], self.sig_x3_f[:, t]
            if self.sig_x_terminal is not None:  # covariance control (i2c.py:453-472)
                sig3m = np.broadcast_to(self.sig_x_terminal, sig3f.shape).copy()
                mu_z, E = self._observe_terminal_lin(mu3f)
                sig_zgx = E @ sig3f @ _T(E)
                sig_zx = E @ sig3f
                mp_inv = np.linalg.inv(sig_zx @ _T(sig_zx))
                sig_z = np.linalg.inv(mp_inv @ (sig_zx @ (sig3f - sig3m) @ _T(sig_zx)) @ _T(mp_inv))
                sig_xi_terminal = sig_z - sig_zgx
                if self.mu_x_terminal is None:
                    G = _T(np.linalg.solve(sig_z, sig_zx))
                    mu3m = mu3f + _mv(G, self.z_term - mu_z)
                else:
                    mu3m = np.broadcast_to(self.mu_x_terminal, mu3f.shape).copy()
            elif self.sig_xi_terminal is not None:  # terminal cost (i2c.py:475-491)
                mu_z, E = self._observe_terminal_lin(mu3f)
                sig_xi_terminal = self.sig_xi_terminal
                sig_z = E @ sig3f @ _T(E) + sig_xi_terminal
                G = _T(np.linalg.solve(_T(sig_z), E @ sig3f))
                mu3m = mu3f + _mv(G, self.z_term - mu_z)
                sig3m = sig3f - G @ E @ sig3f
            else:  # i2c.py:494-497
                mu3m, sig3m = mu3f, sig3f
                sig_xi_terminal = 1e6 * np.eye(nx)
            mu_z3, E = self._observe_terminal_lin(mu3m)  # i2c.py:499-501
            self.mu_z3_m = mu_z3
            self.sig_z3_m = E @ sig3m @ _T(E) + sig_xi_terminal
        else:
            mu3m, sig3m = mu_end, sig_end
        self.mu_x3_m[:, t], self.sig_x3_m[:, t] = mu3m, sig3m

        J = self.J_dyn[:, t]
        mu_m = self.mu_xu1_f[:, t] + _mv(J, mu3m - self.mu_x3_f[:, t])  # i2c.py:515
        sig_m = self.sig_xu1_f[:, t] + J @ (sig3m - self.sig_x3_f[:, t]) @ _T(J)
        self.mu_xu0_m[:, t], self.sig_xu0_m[:, t] = mu_m, sig_m
        sig_xx, sig_ux, sig_uu = sig_m[:, :nx, :nx], sig_m[:, nx:, :nx], sig_m[:, nx:, nx:]
        K = _T(np.linalg.solve(_T(sig_xx), _T(sig_ux)))  # i2c.py:530
        self.K[:, t] = K
        self.k[:, t] = mu_m[:, nx:] - _mv(K, mu_m[:, :nx])
        self.sigK[:, t] = sig_uu - K @ _T(sig_ux)
        # marginal observation: linearised, WITHOUT the x-u cross terms (i2c.py:537-540)
        z, C, D = self._observe_lin(mu_m)
        self.mu_z0_m[:, t] = z
        self.sig_z0_m[:, t] = C @ sig_xx @ _T(C) + D @ sig_uu @ _T(D)
        return mu_m[:, :nx], sig_xx

    # ---- cost: the graph's cubature transform, not the linearised moments (i2c.py:1034-1053, 841-844)
    def calc_cost(self):
        mz, Sz, _, _, _ = self.tf_xu.forward(self.sys.observe, self.mu_xu0_m.reshape(-1, self.d),
                                             self.sig_xu0_m.reshape(-1, self.d, self.d))
        self.mu_z_cost = mz.reshape(self.B, self.H, self.nz)
        self.sig_z_cost = Sz.reshape(self.B, self.H, self.nz, self.nz)
        m, v = self._gaussian_cost(self.mu_z_cost, self.sig_z_cost)
        self.costs_m.append(m.sum(axis=1))
        self.costs_m_var.append(v.sum(axis=1))
        if self._propagate:
            m, v = self._gaussian_cost(self.mu_z0_pf, self.sig_z0_pf)
            self.costs_pf.append(m.sum(axis=1))
            self.costs_pf_var.append(v.sum(axis=1))
        else:
            self.costs_pf.append(-np.ones(self.B))

    # ---- Riccati messages (verification helper of scripts/lqr_compare.py:175) -------------------
    def riccati_sweep(self):
        """I2cGraph._backward_ricatti_msgs (i2c.py:888-893) over I2cCell._backward_ricatti_msgs (:612-678).
        Overwrites K, k, sigK with the Riccati-form controller, as the reference does."""
        nx = self.nx
        I = np.eye(nx)
        nu_b = lam_b = None
        for t in reversed(range(self.H)):
            if nu_b is None:  # i2c.py:615-617
                nu3b = np.linalg.solve(self.sig_x3_m[:, t], self.mu_x3_m[:, t][..., None])[..., 0] - self.nu_x3_f[:, t]
                lam3b = np.linalg.inv(self.sig_x3_m[:, t]) - self.lambda_x3_f[:, t]
            else:
                nu3b, lam3b = nu_b, lam_b
            A, Bm, a = self.A[:, t], self.Bm[:, t], self.a[:, t]
            E, F = self.E[:, t], self.F[:, t]
            mu_u1 = self.mu_xu1_f[:, t, nx:]
            Q = _T(E) @ self.lambda_z1_f[:, t] @ E
            Rug = self.nu_z2_f[:, t]
            nu_u_0 = np.linalg.solve(self.sig_u0_f[:, t], self.mu_u0_f[:, t][..., None])[..., 0]
            lam2f = self.lambda_x2_f[:, t]
            gamma = lam2f @ np.linalg.inv(lam2f + lam3b)
            ALA = _T(A) @ lam3b @ A
            M = np.linalg.inv(self.sig_eta + self.sig_u2_f[:, t]) + lam3b
            ALMLA = _T(A) @ (lam3b @ np.linalg.solve(M, lam3b @ A))
            lam0b = Q + ALA - ALMLA
            AILM = _T(A) @ (I - _T(np.linalg.solve(_T(M), _T(lam3b))))
            nu0b = self.nu_z1_f[:, t] + _mv(AILM, nu3b - _mv(lam3b, a) - _mv(lam3b @ Bm, mu_u1))
            gamma_L = gamma @ lam3b
            igamma = I - gamma
            sig3b = np.linalg.inv(lam3b)
            lam2b = np.linalg.inv(sig3b + self.sig_u2_f[:, t])
            mu_u2 = _mv(Bm, mu_u1)
            nu2b = _mv(lam2b @ sig3b, nu3b) - mu_u2
            psi = gamma_L @ (self.sig_x2_f[:, t] @ (lam2f + np.linalg.inv(sig3b + self.sig_u2_f[:, t])))
            sig_u = self.sig_xu0_m[:, t, nx:, nx:]
            K = -sig_u @ _T(Bm) @ psi @ A
            k = _mv(sig_u, nu_u_0 + Rug + _mv(_T(Bm), _mv(gamma, nu3b) + _mv(igamma, nu2b) - _mv(psi, a)))
            self.K[:, t], self.k[:, t], self.sigK[:, t] = K, k, sig_u
            self.nu_x0_b[:, t], self.lambda_x0_b[:, t] = nu0b, lam0b
            nu_b, lam_b = nu0b, lam0b


__all__ = ["I2cLinearizeOracle", "complex_step_jacobian", "_outer"]
