"""Config 2 of BASELINE.json: fp32 vs fp64 tolerance sweep at the full shapes (double cartpole T=300, pendulum T=200, B=4096).
Runs the fp64 path, the mixed mode (fp64 arithmetic on fp32-stored messages, I2C_F64_F32S) and fp32 arithmetic on the same B
trajectories and reports, per EM iteration, the distribution over the batch of the deviation of the posterior mean from the
fp64 run (fp64 itself is pinned to the reference by the parity tests). The asserted form is tests/test_precision.py.

    python tools/precision_sweep.py [out.json] [B]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from bench_models import CONFIGS, make_env_model, pkg  # noqa: E402


def per_traj(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().flatten(1).max(dim=1).values / b.abs().max()


def sweep(name, B, iters=12):
    cfg = CONFIGS[name]
    model = make_env_model(name)
    T, nu = cfg["T"], model.dim_u
    rng = np.random.default_rng(0)
    x0 = np.asarray(model.x0, float).reshape(1, -1) + 1e-3 * rng.normal(size=(B, model.dim_x))
    mu_u = cfg["mu_u"] * rng.normal(size=(B, T, nu))
    mk = lambda **kw: pkg.BatchedI2c(model, T, cfg["Q"], cfg["R"], cfg["Q"], cfg["alpha"], cfg["tol"], mu_u,  # noqa: E731
                                     cfg["sig_u"] * np.eye(nu), x0=x0, **kw)
    engs = {"fp64": mk(), "fp32_storage": mk(storage_dtype=torch.float32), "fp32_arithmetic": mk(dtype=torch.float32, allow_inexact=True)}
    rows = []
    for it in range(1, iters + 1):
        for e in engs.values():
            e.learn_msgs()
        m64 = engs["fp64"].marginal_state_action()[0]
        c64 = engs["fp64"].costs_m[-1].double()
        row = {"iteration": it}
        for k in ("fp32_storage", "fp32_arithmetic"):
            e = engs[k]
            d = per_traj(e.marginal_state_action()[0], m64)
            d = d[torch.isfinite(d)]
            dc = (e.costs_m[-1].double() - c64).abs() / c64.abs()
            row[k] = {"mean_dev_median": float(d.median()), "mean_dev_p99": float(d.quantile(0.99)), "mean_dev_max": float(d.max()),
                      "within_1e-4": float((d <= 1e-4).double().mean()), "cost_dev_median": float(dc[torch.isfinite(dc)].median()),
                      "failed": len(e.failures())}
        rows.append(row)
        print(name, json.dumps(row))
    return rows


if __name__ == "__main__":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    out = {"B": B, "note": "deviation of the posterior mean from the fp64 run: per trajectory max |diff| / batch-wide max |mean|; "
                           "distribution over the batch, per EM iteration",
           "models": {n: sweep(n, B) for n in ("PendulumKnown", "DoubleCartpoleKnown")}}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "precision_sweep.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
