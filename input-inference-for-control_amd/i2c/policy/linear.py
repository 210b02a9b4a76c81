"""Time-indexed linear-Gaussian policies that consume the solver's (K, k, sigK) output
(reference i2c/policy/linear.py:9-90). Host-side consumers of the hot path's result."""
import numpy as np


class TimeIndexedLinearGaussianPolicy:
    """u_t ~ N(K_t x + k_t, sigK_t + sig_u)."""

    def __init__(self, sig_u, horizon, dim_u, dim_x):
        self.H, self.dim_u, self.dim_x = horizon, dim_u, dim_x
        self.sig_u = np.asarray(sig_u, dtype=float)
        self.zero()

    def zero(self):
        self.K = np.zeros((self.H, self.dim_u, self.dim_x))
        self.k = np.zeros((self.H, self.dim_u))
        self.sigk = np.zeros((self.H, self.dim_u, self.dim_u))

    def write(self, K, k, sigk):
        self.K, self.k, self.sigk = np.array(K, dtype=float), np.array(k, dtype=float), np.array(sigk, dtype=float)

    def mean(self, i, x):
        return self.K[i] @ np.reshape(x, (self.dim_x, 1)) + self.k[i].reshape(self.dim_u, 1)

    def __call__(self, i, x, deterministic=True):
        mu = self.mean(i, x)
        if deterministic:
            return mu
        cov = self.sigk[i] + self.sig_u
        return np.random.multivariate_normal(mu[:, 0], cov, 1).reshape(self.dim_u, 1)


class ExpertTimeIndexedLinearGaussianPolicy(TimeIndexedLinearGaussianPolicy):
    """u_t = k_t + w K_t (x - mu_t) with the pdf-ratio weight w = N(x; mu_t, lam_t^-1) / N(mu_t; ...)
    (`soft=True`) or w = 1; consumes get_local_expert_linear_policy() (reference linear.py:46-90)."""

    def __init__(self, sig_u, horizon, dim_u, dim_x, soft=True):
        self.soft = soft
        super().__init__(sig_u, horizon, dim_u, dim_x)

    def zero(self):
        super().zero()
        self.mu = np.zeros((self.H, self.dim_x))
        self.lam = np.tile(np.eye(self.dim_x), (self.H, 1, 1))

    def write(self, K, k, sigk, mu, lam):
        super().write(K, k, sigk)
        self.mu, self.lam = np.array(mu, dtype=float), np.array(lam, dtype=float)

    def mean(self, i, x):
        dx = np.reshape(x, (self.dim_x, 1)) - self.mu[i].reshape(self.dim_x, 1)
        w = float(np.exp(-0.5 * dx.T @ self.lam[i] @ dx)) if self.soft else 1.0
        return self.k[i].reshape(self.dim_u, 1) + w * (self.K[i] @ dx)
