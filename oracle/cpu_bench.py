"""TEST / BENCH INFRASTRUCTURE ONLY -- times the CPU oracle (oracle/i2c_numpy.py) on the host's cores.

    python oracle/cpu_bench.py [--horizon 200] [--batch 512] [--iters 10] [--workers N]

Each worker process runs the NumPy oracle on its own `batch` pendulum trajectories (trajectories are independent, so
W workers = W x the work); the rate is all cell-iterations divided by the wall time of the slowest worker.
`--batch 512` (default) is the batch-vectorised form; `--batch 1` is the REFERENCE's shape: one trajectory per process, a
Python loop over the T cells with NumPy calls on (d x d) arrays. Workers: one per core, up to 128 -- the count is
in the line. Prints one JSON line. Imports NumPy only (no torch, no GPU).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

from oracle.i2c_numpy import CubatureRule, I2cOracle  # noqa: E402
from oracle.models_numpy import make_model  # noqa: E402


def _work(args):
    rank, T, B, iters = args
    rng = np.random.default_rng(1234 + rank)
    x0 = np.array([np.pi, 0.0]) + 1e-2 * rng.normal(size=(B, 2))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])
    o = I2cOracle(make_model("PendulumKnown"), T, Q, R, Q, 100.0, 0.0, mu_u, 2.0 * np.eye(1), rule=CubatureRule(1, 0, 0), x0=x0)
    o.learn_msgs()  # warm-up: the first iteration is the feed-forward branch
    t0 = time.perf_counter()
    for _ in range(iters):
        o.learn_msgs()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--workers", type=int, default=0, help="0 = one per core, capped at 128")
    a = ap.parse_args()
    cores = os.cpu_count() or 1
    W = a.workers if a.workers > 0 else min(cores, 128)
    t0 = time.perf_counter()
    if W == 1:
        times = [_work((0, a.horizon, a.batch, a.iters))]
    else:
        with mp.get_context("fork").Pool(W) as pool:
            times = pool.map(_work, [(r, a.horizon, a.batch, a.iters) for r in range(W)], chunksize=1)
    wall = max(times)
    print(json.dumps({
        "value": W * a.batch * a.horizon * a.iters / wall,
        "unit": "timestep-messages/s",
        "cores": W,
        "kind": "port",
        "shape": "batch-vectorised" if a.batch > 1 else "reference-shaped: one trajectory per process, Python loop over the cells",
        "sample": f"oracle/i2c_numpy.py ({'batch-vectorised ' if a.batch > 1 else ''}NumPy fp64), pendulum T={a.horizon}: {W} worker processes x {a.batch} "
                  f"trajectories x {a.iters} EM iterations after 1 warm-up; slowest worker {wall:.1f} s, whole leg "
                  f"{time.perf_counter() - t0:.1f} s on a host with {cores} cores",
        "single_worker_rate": a.batch * a.horizon * a.iters / min(times),
    }))


if __name__ == "__main__":
    main()
