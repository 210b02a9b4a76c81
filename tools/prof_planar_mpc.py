"""The planar quadrotor's control step of bench.py (`extra.planar_quadrotor_mpc_H50_B*`: H = 50, two EM iterations per step) on its own,
for a per-kernel profile:   rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/prof_planar_mpc.py [B] [steps]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m = make_env_model("PlanarQuadrotor")
rng = np.random.default_rng(7)
T, n_iter = 50, 2
Q, R = np.diag([1e3, 1e3, 1e3, 1, 1, 1]) / 1e3, np.diag([1e-3, 1e-3])
x0 = np.asarray(m.x0, float).reshape(1, -1) + 1e-2 * rng.normal(size=(B, 6))
mu_u = 0.5 * m.gravity + 1e-2 * rng.normal(size=(B, T, 2))
eng = pkg.BatchedI2c(m, T, Q, R, Q / 1e3, 1.0, 1.0, mu_u, 1e-2 * np.eye(2), x0=x0, keep_zpost=False, keep_xm=False,
                     z_traj=np.broadcast_to(np.concatenate((np.asarray(m.x0, float).reshape(-1), 0.5 * m.gravity * np.ones(2))), (T, 8)))
eng.tau = T - 1
eng.enable_per_cell_alpha()
sig_zeta = 1e-4 * np.eye(8)
y = torch.as_tensor(np.ascontiguousarray(m.measure(x0).T), dtype=torch.float64, device=eng.device)
u = torch.as_tensor(np.ascontiguousarray(mu_u[:, 0, :].T), dtype=torch.float64, device=eng.device)
for _ in range(3):
    eng.mpc_step(n_iter, y, u, sig_zeta)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    eng.mpc_step(n_iter, y, u, sig_zeta)
torch.cuda.synchronize()
print(f"planar quadrotor MPC H={T} n_iter={n_iter} B={B}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per control step "
      f"[{eng.forward_family}/{eng.backward_family}/{eng.backward_schedule}, filter {eng.kernel_family('filter')}], fails {len(eng.failures())}")
