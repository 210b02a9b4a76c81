"""EM iteration time of the other inference rules (Linearize, Gauss-Hermite) next to the cubature rule, pendulum T=200:
python tools/bench_inference.py [B ...]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "input-inference-for-control_amd")]
pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import make_env_model  # noqa: E402


def run(B, inference, T=200, iters=20, **kw):
    m = make_env_model("PendulumKnown")
    rng = np.random.default_rng(0)
    x0 = np.array([np.pi, 0.0]) + 1e-2 * rng.normal(size=(B, 2))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])
    eng = pkg.BatchedI2c(m, T, Q, R, Q, 100.0, 0.0, mu_u, 2.0 * np.eye(1), x0=x0, keep_zpost=False, keep_xm=False, inference=inference, **kw)
    for _ in range(3):
        eng.learn_msgs()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        ev[i][0].record(); eng.forward_sweep(); ev[i][1].record(); eng.backward_sweep(); ev[i][2].record(); eng.maximize()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    f = np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in range(iters)])
    b = np.mean([ev[i][1].elapsed_time(ev[i][2]) for i in range(iters)])
    print(f"pendulum T={T} B={B:6d} {inference:14s} fwd {f:7.3f} bwd {b:7.3f} ms, {wall:7.3f} ms per iteration (stepwise), fails {len(eng.failures())}")


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or [4096]:
        run(B, "cubature")
        run(B, "linearize")
        run(B, "gauss_hermite", gh_degree=3)
