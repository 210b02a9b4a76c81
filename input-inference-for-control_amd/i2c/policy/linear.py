"""Time-indexed linear-Gaussian policies that consume the solver's (K, k, sigK) output -- same class
names, constructor and attributes as the reference (i2c/policy/linear.py:9-90). Called step by step
they run on the host (NumPy); handed to `env.batch_eval` they are recognised and the whole batch of
rollouts runs in one GPU launch (`i2c_rollout`)."""
import numpy as np


class TimeIndexedLinearGaussianPolicy:
    """u_t ~ N(K_t x + k_t, sig_k_t)."""

    device_policy = "linear"

    def __init__(self, sig_u, H, dim_u, dim_x, control_step=1):
        self.H, self.dim_u, self.dim_x = H, dim_u, dim_x
        self.sig_u = np.asarray(sig_u, dtype=float)
        self.control_step = control_step
        self.source = None  # the I2cGraph whose controllers were last written (enables the device path)
        self.init()

    def _blank(self):
        self.K = np.zeros((self.H, self.dim_u, self.dim_x))
        self.k = np.zeros((self.H, self.dim_u))

    def init(self):
        self._blank()
        self.sig_k = np.tile(self.sig_u, (self.H, 1, 1))

    def zero(self):
        self._blank()
        self.sig_k = np.zeros((self.H, self.dim_u, self.dim_u))
        self.source = None

    def write(self, K, k, sig_k):
        self.K[...] = K
        self.k[...] = k
        self.sig_k[...] = sig_k

    def __call__(self, i, x, deterministic=True):
        assert i < self.H
        if i % self.control_step == 0:
            mean = self.K[i] @ x + self.k[i][:, None]
            self.u = mean if deterministic else np.random.multivariate_normal(mean[:, 0], self.sig_k[i], 1)
        return self.u


class ExpertTimeIndexedLinearGaussianPolicy:
    """u_t = k_t + p K_t (x - mu_t) with p = exp(-e) (soft) or [|e| < 3] (hard), e = (x-mu)' lam (x-mu) / 2;
    consumes get_local_expert_linear_policy()."""

    hard_exp_threshold = 3.0

    def __init__(self, sig_u, H, dim_u, dim_x, soft=True):
        self.H, self.dim_u, self.dim_x = H, dim_u, dim_x
        self.sig_u = np.asarray(sig_u, dtype=float)
        self.soft = soft
        self.source = None
        self.init()

    @property
    def device_policy(self):
        return "expert_soft" if self.soft else "expert_hard"

    def init(self):
        self.K = np.zeros((self.H, self.dim_u, self.dim_x))
        self.k = np.zeros((self.H, self.dim_u))
        self.sig_k = np.tile(self.sig_u, (self.H, 1, 1))
        self.mu = np.zeros((self.H, self.dim_x))
        self.lam = np.ones((self.H, self.dim_x, self.dim_x))

    def zero(self):
        self.init()
        self.source = None

    def write(self, K, k, sig_k, mu, lam):
        self.K[...] = K
        self.k[...] = k
        self.sig_k[...] = sig_k
        self.mu[...] = mu
        self.lam[...] = lam

    def __call__(self, i, x, deterministic=True):
        assert i < self.H
        dist = x - self.mu[i][:, None]
        e = (0.5 * dist.T @ self.lam[i] @ dist).item()
        p = np.exp(-e) if self.soft else float(abs(e) < self.hard_exp_threshold)
        mean = self.k[i][:, None] + p * (self.K[i] @ dist)
        if deterministic:
            return mean.reshape(self.dim_u, 1)
        noise = np.random.multivariate_normal(np.zeros(self.dim_u), self.sig_k[i].reshape(self.dim_u, self.dim_u), 1)
        return mean + noise.reshape(self.dim_u, 1)
