"""Closed-loop propagation (i2c.py:700-748) of the 12-state quadrotor, T = 50: the quad kernel (four trajectories per wavefront,
v_mfma_f64_4x4x4_4b) beside the group kernel (16 lanes per trajectory) on the same engine state:
    python tools/bench_prop12.py [B ...]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "input-inference-for-control_amd")]
pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import make_env_model  # noqa: E402


def run(B, T=50, K=20):
    m = make_env_model("Quadrotor12")
    rng = np.random.default_rng(7)
    Q, R = np.diag([10.0] * 3 + [1.0] * 3 + [0.1] * 6), 1e-2 * np.eye(4)
    x0 = 1e-2 * rng.normal(size=(B, 12))
    mu_u = 0.25 * m.gravity + 1e-2 * rng.normal(size=(B, T, 4))
    out = {}
    for lanes in (0, 16):
        eng = pkg.BatchedI2c(m, T, Q, R, Q / 10.0, 0.02, 1.0, mu_u, 1e-2 * np.eye(4), x0=x0, keep_zpost=False, group_lanes=lanes,
                             z_traj=np.broadcast_to(np.concatenate((m.zg_term.reshape(-1), 0.25 * m.gravity * np.ones(4))), (T, 16)))
        eng.use_expert_controller = False
        for _ in range(2):
            eng.learn_msgs()
        for _ in range(3):
            eng.propagate()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(K):
            eng.propagate()
        ev[1].record()
        torch.cuda.synchronize()
        out[lanes] = (ev[0].elapsed_time(ev[1]) / K, eng.kernel_family("propagate"), eng.prop.clone(), len(eng.failures()))
    d = (out[0][2] - out[16][2]).abs().max().item() / out[16][2].abs().max().item()
    print(f"Quadrotor12 propagate T={T} B={B:6d}: {out[0][1]} {out[0][0]:.3f} ms, {out[16][1]} {out[16][0]:.3f} ms; "
          f"max rel diff {d:.2e}; fails {out[0][3]}/{out[16][3]}")


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 8192]:
        run(B)
