"""Closed-loop MPC step time (PartiallyObservedMpcPolicy protocol, scripts/mpc_state_est/mpc_quad.py:624-650):
filter + n_iter x (forward + backward + prior update) + first action + horizon shift, B loops at once.
    python tools/bench_mpc.py [B ...]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "input-inference-for-control_amd")]
pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.exp_types import CubatureQuadrature  # noqa: E402
from i2c.i2c import I2cGraph  # noqa: E402
from i2c.known_models import make_env_model  # noqa: E402
from i2c.policy.mpc import PartiallyObservedMpcPolicy  # noqa: E402


def run(B, T=50, n_iter=1, steps=30):
    model = make_env_model("PlanarQuadrotor")
    nu = model.dim_u
    Q, R = np.diag([1e3, 1e3, 1e3, 1, 1, 1]) / 1e3, np.diag([1e-3, 1e-3])
    mu_u = 0.5 * model.gravity * np.ones((T, nu))
    model.sig_zeta = 1e-4 * np.eye(8)
    g = I2cGraph(model, T, Q, R, Q, 1.0, 1.0, mu_u, 1e-2 * np.eye(nu), None, None, CubatureQuadrature(1, 0, 0), batch=B)
    pol = PartiallyObservedMpcPolicy(g, n_iter, 1e-2 * np.eye(nu))
    pol.set_control(False)
    y = np.zeros((B, 8))
    u = np.zeros((B, nu))
    for i in range(3):
        u = pol(i, y, u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 3 + steps):
        u = pol(i, y, np.asarray(u).reshape(B, nu))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"quadrotor MPC  B={B:6d} T={T} n_iter={n_iter}: {dt * 1e3:8.3f} ms / control step, {B / dt:10.3e} closed-loop steps/s")


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or [1, 1024, 8192]:
        run(B)
