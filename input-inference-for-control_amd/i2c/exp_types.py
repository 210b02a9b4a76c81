"""Experiment hyper-parameter types with the reference's names and fields
(reference i2c/exp_types.py): configs written for the reference (e.g.
scripts/experiments/pendulum_known_quad.py) construct these unchanged.

The rule objects are plain parameter carriers here; the sigma-point arithmetic itself runs in
the HIP kernels (csrc/i2c_cell.hpp), which receive (alpha, beta, kappa) through the C ABI.
``pts`` / ``weights`` are kept because host-side callers (QuadratureInference) use them.
"""
import dataclasses
from typing import Any

import numpy as np


@dataclasses.dataclass
class GaussianI2c:
    """Bundle of solver hyper-parameters, field for field as the reference's dataclass."""

    inference: Any
    alpha: float
    alpha_update_tol: float
    Q: Any
    Qf: Any
    R: Any
    mu_u: Any
    sig_u: Any
    mu_x_term: Any
    sig_x_term: Any


@dataclasses.dataclass
class Linearize:
    """EKF-style inference (no parameters). Not part of the GPU hot path."""


@dataclasses.dataclass
class CubatureQuadrature:
    """Scaled symmetric sigma-point rule with 2 dim + 1 points."""

    alpha: float
    beta: float
    kappa: float

    @staticmethod
    def pts(dim):
        unit = np.eye(dim)
        return np.vstack((np.zeros((1, dim)), unit, -unit))

    def weights(self, dim):
        if not self.alpha > 0:
            raise AssertionError("alpha must be positive")
        spread = self.alpha ** 2 * (dim + self.kappa)  # = dim + lambda
        w = np.full(2 * dim + 1, 0.5 / spread)
        w[0] = (spread - dim) / spread
        w_cov = w.copy()
        w_cov[0] += 1.0 - self.alpha ** 2 + self.beta
        return np.sqrt(spread), w, w_cov

    def as_tuple(self):
        return float(self.alpha), float(self.beta), float(self.kappa)


@dataclasses.dataclass
class GaussHermiteQuadrature:
    """Tensor-grid Gauss-Hermite rule with degree ** dim points (runs on the device up to degree 8)."""

    degree: int

    def __post_init__(self):
        if self.degree < 1:
            raise AssertionError("degree must be >= 1")
        self.gh_pts, self.gh_weights = np.polynomial.hermite.hermgauss(self.degree)

    def _grid(self, values, dim):
        mesh = np.meshgrid(*([values] * dim))
        return np.stack([g.reshape(-1) for g in mesh], axis=1)

    def pts(self, dim):
        return self._grid(self.gh_pts, dim)

    def weights(self, dim):
        w = self._grid(self.gh_weights, dim).prod(axis=1) / np.pi ** (dim / 2.0)
        return np.sqrt(2.0), w, w
