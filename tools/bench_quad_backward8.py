"""Round 6: the d <= 8 quad backward sweep (backward_quad8_body) against the lane schedules, sweep by sweep.
    python tools/bench_quad_backward8.py [model ...] [B ...] [mpc]
For every (model, B): forward / backward sweep (HIP events, 20 repetitions) and one EM iteration (i2c_learn, wall clock) with
  default     what the resolver picks (quad forward inside the model's window; chunked backward schedule, its walk pass on the quad
              walker up to a few hundred trajectories, on the lane walker beyond)
  quad        group_lanes = 64: quad forward + quad backward (the fused walk of four trajectories per wavefront)
  quad-chunk  group_lanes = 64 + backward_mode = "chunked": the chunked schedule with the quad WALKER (four trajectories per wavefront
              and chunk; compose / stitch / reduce stay lane kernels)
  lane-chunk  quad forward sweep (LANES_QUAD) + the chunked schedule on the lane walker, asked for by name
  lane-fused  the lane kernels' fused walk behind the default forward sweep (same bytes as the quad walk, one trajectory per lane)
`mpc`: the planar-quadrotor control step of bench.py (H = 50, two EM iterations per step) at B = 1024 / 8192, default vs quad."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "input-inference-for-control_amd"), os.path.join(ROOT, "tools")]
pkg = importlib.import_module("input-inference-for-control_amd")
from bench_models import CONFIGS  # noqa: E402
from i2c.known_models import make_env_model  # noqa: E402


def engine(name, B, **kw):
    cfg = CONFIGS[name]
    model = make_env_model(name)
    T, nu = cfg["T"], model.dim_u
    rng = np.random.default_rng(0)
    x0 = np.asarray(model.x0, float).reshape(1, -1) + 1e-3 * rng.normal(size=(B, model.dim_x))
    base = {"PlanarQuadrotor": 0.5}.get(name, 0.0) * getattr(model, "gravity", 0.0)
    mu_u = base + cfg["mu_u"] * rng.normal(size=(B, T, nu))
    return pkg.BatchedI2c(model, T, cfg["Q"], cfg["R"], cfg["Q"], cfg["alpha"], cfg["tol"], mu_u, cfg["sig_u"] * np.eye(nu), x0=x0,
                          keep_zpost=False, keep_xm=False, **kw)


def sweeps(name, B, reps=20):
    rows = []
    for tag, kw in (("default", {}), ("lane-chunk", dict(group_lanes=pkg._native.LANES_QUAD, backward_mode="chunked")), ("quad", dict(group_lanes=64)),
                    ("quad-chunk", dict(group_lanes=64, backward_mode="chunked")), ("lane-fused", dict(backward_mode="fused"))):
        eng = engine(name, B, **kw)
        eng.learn(3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            eng.forward_sweep()
        ev[1].record()
        for _ in range(reps):
            eng.backward_sweep()
        ev[2].record()
        torch.cuda.synchronize()
        fwd, bwd = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.learn(reps)
        torch.cuda.synchronize()
        it = (time.perf_counter() - t0) / reps * 1e3
        d = eng.dims
        gb = (d.e_fwd + d.e_post) * 8 * B * eng.H / bwd / 1e6
        rows.append((tag, fwd, bwd, it))
        print(f"{name:20s} B={B:6d} T={eng.H:3d} {tag:10s} [{eng.forward_family:5s}/{eng.backward_family:5s} {eng.backward_schedule:8s}] "
              f"fwd {fwd:7.3f}  bwd {bwd:7.3f} ms ({gb:7.1f} GB/s algorithmic)  EM iteration {it:7.3f} ms  fails {len(eng.failures())}", flush=True)
        del eng
    return rows


def mpc(B, K=20):
    m = make_env_model("PlanarQuadrotor")
    rng = np.random.default_rng(7)
    for tag, kw in (("default", {}), ("quad", dict(group_lanes=64)), ("quad-chunk", dict(group_lanes=64, backward_mode="chunked"))):
        T, n_iter = 50, 2
        Q, R = np.diag([1e3, 1e3, 1e3, 1, 1, 1]) / 1e3, np.diag([1e-3, 1e-3])
        x0 = np.asarray(m.x0, float).reshape(1, -1) + 1e-2 * rng.normal(size=(B, 6))
        mu_u = 0.5 * m.gravity + 1e-2 * rng.normal(size=(B, T, 2))
        eng = pkg.BatchedI2c(m, T, Q, R, Q / 1e3, 1.0, 1.0, mu_u, 1e-2 * np.eye(2), x0=x0, keep_zpost=False, keep_xm=False,
                             z_traj=np.broadcast_to(np.concatenate((np.asarray(m.x0, float).reshape(-1), 0.5 * m.gravity * np.ones(2))), (T, 8)), **kw)
        eng.tau = T - 1
        eng.enable_per_cell_alpha()
        sig_zeta = 1e-4 * np.eye(8)
        y = torch.as_tensor(np.ascontiguousarray(m.measure(x0).T), dtype=torch.float64, device="cuda")
        u = torch.as_tensor(np.ascontiguousarray(mu_u[:, 0, :].T), dtype=torch.float64, device="cuda")
        for _ in range(3):
            eng.mpc_step(n_iter, y, u, sig_zeta)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            eng.mpc_step(n_iter, y, u, sig_zeta)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        print(f"planar quadrotor MPC H=50 n_iter=2 B={B:5d} {tag:8s} [{eng.forward_family}/{eng.backward_family} {eng.backward_schedule}] {ms:7.3f} ms per control step  fails {len(eng.failures())}", flush=True)
        del eng


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in CONFIGS] or ["DoubleCartpoleKnown", "PlanarQuadrotor", "CartpoleKnown", "PendulumKnown"]
    Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 8192, 16384]
    for n in names:
        for B in Bs:
            sweeps(n, B)
    if "mpc" in sys.argv[1:]:
        for B in (1024, 8192):
            mpc(B)
