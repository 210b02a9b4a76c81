"""Headline benchmark: i2c EM iterations/s and timestep-messages/s, pendulum T=200, B=4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--horizon T] [--dtype f64|f32]

A "step" is ONE EM iteration (I2cGraph.learn_msgs: forward sweep, backward sweep, temperature
M-step) over the whole local batch; inputs are resident in HBM before the timed region. One
cell-iteration ("timestep-message") = one (trajectory, timestep) cell through forward message +
backward message + M-step statistics. N > 1: one process per GPU (torchrun),
the batch axis is sharded with no collective inside the EM loop (weak scaling: B per GPU is
fixed); the single RCCL all-gather of the final controllers runs after the timed steps and is
reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "input-inference-for-control_amd"
for p in (ROOT, os.path.join(ROOT, PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)


def synthetic_pendulum_inputs(B, T, rank=0):
    """x0_b = [pi, 0] + 1e-2 eps_b (default_rng(1234 + rank)); mu_u[b] = 1e-2 randn(T, 1)."""
    rng = np.random.default_rng(1234 + rank)
    x0 = np.array([np.pi, 0.0]) + 1e-2 * rng.normal(size=(B, 2))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    if rank == 0:
        x0[0] = [np.pi, 0.0]  # trajectory 0 = the reference's own problem
    return x0, mu_u


def make_engine(pkg, B, T, dtype, device, rank=0, backward_mode="auto"):
    from i2c.known_models import make_env_model

    x0, mu_u = synthetic_pendulum_inputs(B, T, rank)
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])  # scripts/experiments/pendulum_known_quad.py:22-33
    return pkg.BatchedI2c(make_env_model("PendulumKnown"), T, Q, R, Q, 100.0, 0.0, mu_u, 2.0 * np.eye(1), x0=x0,
                          dtype=dtype, device=device, keep_zpost=False, keep_xm=False, backward_mode=backward_mode)


def timed_iterations(eng, K, sync):
    """K EM iterations with HIP events around each sweep (recorded on the launch stream).
    Returns (wall seconds, mean ms of forward / backward / mstep)."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    sync()
    t0 = time.perf_counter()
    for i in range(K):
        eng.em_iter += 1
        ev[i][0].record()
        eng.forward_sweep()
        ev[i][1].record()
        eng.backward_sweep()
        ev[i][2].record()
        eng.maximize()
        ev[i][3].record()
    sync()
    elapsed = time.perf_counter() - t0
    ms = [float(np.mean([ev[i][j].elapsed_time(ev[i][j + 1]) for i in range(K)])) for j in range(3)]
    return elapsed, ms


def saturated_leg(pkg, T, dtype, device, el, wbytes, B=131072, K=6):
    """Same kernels at a batch that fills the chip (2 wavefronts per SIMD): shows the HBM-bound
    regime the headline batch (64 wavefronts on 1024 SIMDs) cannot reach."""
    eng = make_engine(pkg, B, T, dtype, device, rank=7)
    for _ in range(2):
        eng.learn_msgs()
    elapsed, ms = timed_iterations(eng, K, lambda: torch.cuda.synchronize(device))
    cells = B * T
    schedule = eng.backward_schedule
    del eng
    # for scale: the streaming-copy rate of this device (read + write of a 1 GiB fp64 tensor), measured live
    x = torch.empty(1024 ** 3 // 8, dtype=torch.float64, device=device).normal_()
    y = torch.empty_like(x)
    y.copy_(x)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        y.copy_(x)
    ev1.record()
    torch.cuda.synchronize(device)
    copy_gbps = 2 * x.numel() * 8 / (ev0.elapsed_time(ev1) / 10 * 1e-3) / 1e9
    return {
        "batch": B,
        "device_memcpy_GBps": copy_gbps,
        "value": cells * K / elapsed,
        "unit": "timestep-messages/s",
        "ms_per_step": elapsed / K * 1e3,
        "backward": schedule,
        "kernel_ms": {"forward_sweep": ms[0], "backward_sweep": ms[1], "mstep": ms[2]},
        "forward_GBps": el["forward"] * wbytes * cells / (ms[0] * 1e-3) / 1e9,
        "backward_GBps": el["backward"] * wbytes * cells / (ms[1] * 1e-3) / 1e9,
        "whole_iteration_GBps": el["total"] * wbytes * cells / (elapsed / K) / 1e9,
        "whole_iteration_frac_of_peak": el["total"] * wbytes * cells / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
    }


def algorithmic_elements(d, nx, nu):
    """elements moved per cell-iteration, by sweep."""
    s = lambda n: n * (n + 1) // 2  # noqa: E731
    fwd_read = d + s(d) + nu * nx
    fwd_write = d + s(d) + nx + s(nx) + d * nx
    bwd_write = d + s(d) + nu * nx + nu + s(nu)
    return dict(forward=fwd_read + fwd_write, backward=fwd_write + bwd_write, total=fwd_read + 2 * fwd_write + bwd_write)


def measured_traffic(B, T, dtype, kernel):
    """HBM bytes per launch from the committed PMC passes (tools/pmc_summary.py), if one matches."""
    path = os.path.join(ROOT, "profiles", f"r1_B{B}_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if (d["B"], d["T"], d["dtype"]) != (B, T, dtype) or kernel not in d["kernels"]:
        return None
    return d["kernels"][kernel]["hbm_bytes_per_launch"]


def cpu_baseline(T):
    """The oracle (NumPy restatement, batch-vectorised) timed on THIS host's cores, as a child process that
    imports NumPy only (oracle/cpu_bench.py): one worker per core (capped at 32), a bounded sample."""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--horizon", str(T), "--batch", "512", "--iters", "10"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories PER GPU")
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the extra B=131072 leg")
    ap.add_argument("--backward", default="auto", choices=["auto", "two_pass", "fused", "chunked"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:  # launched by torch.distributed.run (also with N = 1)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)  # RCCL on ROCm

    pkg = importlib.import_module(PKG)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    B, T = args.batch, args.horizon
    eng = make_engine(pkg, B, T, dtype, device, rank, args.backward)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        eng.learn_msgs()

    # ---- per-kernel timing: K EM iterations stepped from Python with HIP events around each sweep -----
    K = args.steps
    stepwise_s, (fwd_ms, bwd_ms, mst_ms) = timed_iterations(eng, K, barrier)

    # ---- THE timed region: exactly K EM iterations, enqueued by ONE i2c_learn call (the way a caller runs
    # N iterations: no Python between sweeps), bracketed by barrier + synchronize on both sides ---------
    barrier()
    t0 = time.perf_counter()
    eng.learn(K)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    n_fail = len(eng.failures())

    # ---- the one collective of the job: all-gather of the final controllers -------
    allgather_ms = None
    if dist is not None:
        gather_policy = importlib.import_module(PKG + ".dist").gather_policy

        gather_policy(eng)  # first call pays RCCL's lazy channel setup; time the steady-state exchange
        torch.cuda.synchronize(device)
        dist.barrier()
        t1 = time.perf_counter()
        gathered = gather_policy(eng)
        torch.cuda.synchronize(device)
        allgather_ms = (time.perf_counter() - t1) * 1e3
        assert gathered["K"].shape[0] == B * world

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    cells = B * T * world
    wbytes = 8 if args.dtype == "f64" else 4
    el = algorithmic_elements(eng.d, eng.nx, eng.nu)
    fwd_bytes = el["forward"] * wbytes * B * T  # per launch of the forward-sweep kernel
    achieved = fwd_bytes / (fwd_ms * 1e-3) / 1e9
    out = {
        "metric": "i2c timestep-messages/sec (cell-iterations/s = B*T*EM-iters/s), pendulum T=200 B=4096 per GPU",
        "value": cells * K / elapsed,
        "unit": "timestep-messages/s",
        "em_iters_per_sec": K / elapsed,
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"pendulum_known_quad cubature i2c (nx=2, nu=1, nz=4), T={T}, B={B} trajectories per GPU, "
                        f"global batch {B * world}, CubatureQuadrature(1,0,0), alpha0=100, tol=0",
            "batch_per_gpu": B,
            "horizon": T,
            "parallelism": f"batch-sharded x{world}, no collective in the EM loop",
        },
        "kernel_ms": {"forward_sweep": fwd_ms, "backward_sweep": bwd_ms, "mstep": mst_ms},
        "backward": eng.backward_schedule,
        "ms_per_step_stepped_from_python": stepwise_s / K * 1e3,
        "failed_trajectories": n_fail,
        "roofline": {
            "kernel": "k_forward (forward sweep, the dominant kernel)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(B, T, args.dtype, "k_forward"),
            "traffic_note": "bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                            "(FETCH_SIZE x2, gfx950 correction; profiles/r1_B*_pmc_traffic.json); algorithmic = "
                            + str(fwd_bytes),
            "algorithmic_bytes_per_cell": {k: v * wbytes for k, v in el.items()},
            "whole_iteration_GBps": el["total"] * wbytes * B * T / (elapsed / K) / 1e9,
            "note": "instruction-issue-bound at B=4096 per GPU: 64 lone wavefronts (1024 SIMDs), each issuing ~88 % of its cycles "
                    "(profiles/r1_k_forward_sq_counters.json); HBM-bound from B ~ 32768 (saturated_batch); see DESIGN.md section 6",
        },
        "final_allgather_ms": allgather_ms,
    }
    if not args.no_saturated and world == 1:
        out["saturated_batch"] = saturated_leg(pkg, T, dtype, device, el, wbytes)
    if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(T)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
