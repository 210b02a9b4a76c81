"""Host-side `QuadratureInference` with the reference's interface
(reference i2c/inference/quadrature.py:7-58): Gaussian -> Gaussian push-through of an arbitrary
Python callable through a sigma-point rule.

This class exists because callers use it *directly* (the CKF inside policy/mpc.py, scripts/
mpc_state_est/mpc_quad.py:129-152). It is NOT what I2cGraph runs: the solver's push-throughs are
`sp_transform` in csrc/i2c_cell.hpp on the GPU, and the batched CKF is `i2c_ckf_filter`.
"""
import logging

import numpy as np

from ..exp_types import CubatureQuadrature, GaussHermiteQuadrature


class QuadratureInference:
    def __init__(self, params, dim):
        if not isinstance(params, (CubatureQuadrature, GaussHermiteQuadrature)):
            raise AssertionError("params must be a quadrature rule")
        self.dim = dim
        self.base_pts = params.pts(dim)
        self.sf, self.weights_mu, self.weights_sig = params.weights(dim)
        self.n_points = self.base_pts.shape[0]
        self.x_pts = self.y_pts = self.m_y = self.sig_y = self.sig_xy = self.sig_noise = None

    def get_x_pts(self, m_x, sig_x):
        centre = np.reshape(m_x, (1, self.dim))
        try:
            factor = np.linalg.cholesky(sig_x)
        except np.linalg.LinAlgError:
            logging.exception(f"Bad Cholesky\nCov:\n{sig_x}\nEigvals:\n{np.linalg.eigvalsh(sig_x)}")
            raise
        return centre + self.base_pts @ (self.sf * factor).T

    def _moments(self, m_x, x_pts, y_pts):
        w = self.weights_sig
        mean = w @ y_pts
        cov = (y_pts * w[:, None]).T @ y_pts - np.outer(mean, mean)
        cross = (x_pts * w[:, None]).T @ y_pts - np.outer(np.reshape(m_x, -1), mean)
        return mean.reshape(1, -1), cov, cross

    def forward_pts(self, f, m_x, x_pts):
        y_pts = f(x_pts)
        m_y, sig_y, sig_xy = self._moments(m_x, x_pts, y_pts)
        return y_pts, m_y, sig_y, sig_xy

    def forward(self, f, m_x, sig_x):
        self.x_pts = self.get_x_pts(m_x, sig_x)
        self.y_pts, self.m_y, self.sig_y, self.sig_xy = self.forward_pts(f, m_x, self.x_pts)
        return self.m_y.T, self.sig_y

    def forward_gaussian(self, f, m_x, sig_x):
        self.x_pts = self.get_x_pts(m_x, sig_x)
        self.y_pts, noise = f(self.x_pts)
        self.m_y, self.sig_y, self.sig_xy = self._moments(m_x, self.x_pts, self.y_pts)
        self.sig_noise = np.tensordot(self.weights_sig, noise, axes=(0, 0))
        return self.m_y.T, self.sig_y, self.sig_noise
