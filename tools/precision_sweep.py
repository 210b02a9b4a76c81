"""Config 2 of BASELINE.json: fp32 vs fp64 tolerance sweep (double cartpole T=300, pendulum T=200).
Runs both precisions on the same B trajectories and reports, per EM iteration, the max-norm relative
deviation of the fp32 run from the fp64 run (fp64 itself is pinned to the reference by the parity tests).

    python tools/precision_sweep.py [out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from bench_models import CONFIGS, make_env_model, pkg  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max())


def sweep(name, B=64, iters=12):
    cfg = CONFIGS[name]
    model = make_env_model(name)
    T, nu = cfg["T"], model.dim_u
    rng = np.random.default_rng(0)
    x0 = np.asarray(model.x0, float).reshape(1, -1) + 1e-3 * rng.normal(size=(B, model.dim_x))
    mu_u = cfg["mu_u"] * rng.normal(size=(B, T, nu))
    engs = {}
    for dt in (torch.float64, torch.float32):
        engs[dt] = pkg.BatchedI2c(model, T, cfg["Q"], cfg["R"], cfg["Q"], cfg["alpha"], cfg["tol"], mu_u,
                                  cfg["sig_u"] * np.eye(nu), x0=x0, dtype=dt, allow_inexact=True)
    rows = []
    for it in range(1, iters + 1):
        for e in engs.values():
            e.learn_msgs()
        e64, e32 = engs[torch.float64], engs[torch.float32]
        m64, s64 = e64.marginal_state_action()
        m32, s32 = e32.marginal_state_action()
        K64, k64, _ = e64.local_linear_policy()
        K32, k32, _ = e32.local_linear_policy()
        ok = torch.isfinite(m32).all(dim=(1, 2))
        rows.append(dict(iteration=it, mean=rel(m32[ok], m64[ok]), cov=rel(s32[ok], s64[ok]), K=rel(K32[ok], K64[ok]),
                         k=rel(k32[ok], k64[ok]), alpha=rel(e32.alpha[ok], e64.alpha[ok]),
                         fp32_failed=len(e32.failures()), fp64_failed=len(e64.failures())))
        print(name, rows[-1])
    return rows


if __name__ == "__main__":
    out = {n: sweep(n) for n in ("PendulumKnown", "DoubleCartpoleKnown")}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "precision_sweep.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
