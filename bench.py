"""Headline benchmark: i2c EM iterations/s and timestep-messages/s, pendulum T=200, B=4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--horizon T]

`--gpus N` with N > 1 starts its own N workers (one process per GPU through torch.distributed.run on 127.0.0.1) unless it
is already running under torchrun (RANK / WORLD_SIZE in the environment, the way the driver launches it).

A "step" is ONE EM iteration (I2cGraph.learn_msgs: forward sweep, backward sweep, temperature
M-step) over the whole local batch; inputs are resident in HBM before the timed region. One
cell-iteration ("timestep-message") = one (trajectory, timestep) cell through forward message +
backward message + M-step statistics. N > 1: one process per GPU (torchrun),
the batch axis is sharded with no collective inside the EM loop (weak scaling: B per GPU is
fixed); the single RCCL all-gather of the final controllers runs after the timed steps and is
reported separately. Two STRONG-scaling legs (global batch 4096 and 65536 split over the ranks) and, at
N = 1, legs for the other BASELINE configs (double cartpole T=300 B=4096, 12-state quadrotor MPC step H=50 B=8192,
pendulum covariance control T=100 B=8192) ride on the same JSON line.

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


def make_engine(pkg, B, T, dtype, device, rank=0, backward_mode="auto", lib=None, group_lanes=0, storage_dtype=None):
    from i2c.known_models import make_env_model

    x0, mu_u = synthetic_pendulum_inputs(B, T, rank)
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])  # scripts/experiments/pendulum_known_quad.py:22-33
    return pkg.BatchedI2c(make_env_model("PendulumKnown"), T, Q, R, Q, 100.0, 0.0, mu_u, 2.0 * np.eye(1), x0=x0,
                          dtype=dtype, device=device, keep_zpost=False, keep_xm=False, backward_mode=backward_mode, lib=lib,
                          group_lanes=group_lanes, storage_dtype=storage_dtype)


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


def saturated_leg(pkg, T, dtype, device, el, wbytes, B=131072, K=6, storage_dtype=None):
    """Same kernels at a batch that fills the chip (2 wavefronts per SIMD): shows the HBM-bound
    regime the headline batch (64 wavefronts on 1024 SIMDs) cannot reach. With storage_dtype=torch.float32: the mixed mode
    (fp64 arithmetic on fp32-stored messages, half the bytes; deviation from fp64 bounded by tests/test_precision.py)."""
    eng = make_engine(pkg, B, T, dtype, device, rank=7, storage_dtype=storage_dtype)
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
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_B{B}_pmc_traffic.json") for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)),
                os.path.join(ROOT, "profiles", f"r1_B{B}_pmc_traffic.json"))
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if (d["B"], d["T"], d.get("dtype", "f64")) != (B, T, dtype) or kernel not in d["kernels"]:
        return None
    return d["kernels"][kernel]["hbm_bytes_per_launch"]


FP64_PEAK_TFLOPS = 78.6  # MI355X vector / matrix fp64: 1024 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz (the matrix rate is the same:
#                          one v_mfma_f64_16x16x4 = 2048 flop in 64 clocks, measured, tools/micro/mfma_f64.hip)


def issue_roofline(profile, kernel_prefix, cells_per_wave, lanes_per_trajectory, kernel_ms, cells, useful=None):
    """The instruction-issue side of a kernel that is NOT memory-bound, from the committed SQ counter passes
    (tools/sq_counters.sh -> profiles/<profile>_sq_counters.json): how busy the SIMD's fp64 pipe is while a wave is resident
    (4 clocks per vector instruction of a wave64, 16 clocks per 512-flop unit of an fp64 matrix instruction), how many of the chip's
    1024 SIMDs hold a wave at all, and the executed fp64 flops against the chip's peak at the live kernel time."""
    def latest(name):  # "rN_<name>": the newest round that committed this profile
        return next((f"r{r}_{name}" for r in (6, 5, 4, 3, 2) if os.path.exists(os.path.join(ROOT, "profiles", f"r{r}_{name}_sq_counters.json"))), "r4_" + name)

    profile = latest(profile)
    if useful is not None:
        useful = (latest(useful[0]), useful[1])
    path = os.path.join(ROOT, "profiles", profile + "_sq_counters.json")
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    k = next((v for name, v in ks.items() if name.startswith(kernel_prefix) and "issue" in v), None)
    if k is None:
        return None
    i = k["issue"]
    flops_cell = (i["fp64_vector_flops_per_lane_per_wave"] * lanes_per_trajectory + i["mfma_flops_per_wave"] / (64 // lanes_per_trajectory)) / cells_per_wave
    executed = flops_cell
    if useful is not None:  # a multi-lane kernel repeats work on the lanes of a trajectory: the USEFUL count is the one-lane kernel's
        up = os.path.join(ROOT, "profiles", useful[0] + "_sq_counters.json")
        uk = next((v for name, v in (json.load(open(up))["kernels"] if os.path.exists(up) else {}).items() if name.startswith(useful[1]) and "issue" in v), None)
        if uk is not None:
            flops_cell = uk["issue"]["fp64_vector_flops_per_lane_per_wave"] / cells_per_wave
    return {
        "source": "profiles/" + profile + "_sq_counters.json",
        "vector_insts_per_cell": i["vector_insts_per_wave"] / cells_per_wave,
        "mfma_insts_per_cell": i["mfma_insts_per_wave"] / cells_per_wave,
        "wave_clocks_per_cell": i["wave_clocks"] / cells_per_wave,
        "fp64_issue_frac": i["fp64_issue_frac"],
        "simds_occupied": i["simds_occupied"],
        "useful_flops_per_cell": flops_cell,
        "executed_flops_per_cell": executed,
        "useful_flops_note": "fp64 flops per trajectory-cell (2 x fma + add + mul, + matrix-instruction flops). useful = what the one-lane-per-"
                             "trajectory kernel of the same model executes (nothing redundant); executed = this kernel, all lanes of the "
                             "trajectory (a multi-lane kernel repeats the pivot algebra on its sixteen lanes and multiplies zero padding)",
        "frac_of_fp64_peak": flops_cell * cells / (kernel_ms * 1e-3) / (FP64_PEAK_TFLOPS * 1e12),
        "peak_TFLOPs": FP64_PEAK_TFLOPS,
    }


def cpu_baseline(T):
    """The oracle (NumPy restatement) timed on THIS host's cores, as child processes that import NumPy only
    (oracle/cpu_bench.py), one worker per core up to 128, bounded samples. Two shapes: batch-vectorised (the strongest CPU form
    of this restatement: 512 trajectories per worker) and, as `reference_shaped`, the reference's own shape -- ONE trajectory per
    process, a Python loop over the T cells."""
    import subprocess

    def run(*extra):
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--horizon", str(T)] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    cores = os.cpu_count() or 1
    runs = [run("--batch", "512", "--iters", "4", "--workers", str(w)) for w in sorted({min(cores, 32), min(cores, 128)})]
    out = dict(max(runs, key=lambda r: r["value"]))  # the strongest CPU form is the stated baseline
    out["by_workers"] = {str(r["cores"]): r["value"] for r in runs}  # NumPy workers contend for memory bandwidth: more is not faster
    out["reference_shaped"] = run("--batch", "1", "--iters", "5")
    return out


def launch_workers(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N workers (one per GPU) as a child `torch.distributed.run` and relay
    rank 0's JSON line. This process never touches the GPU (no torch.cuda call), so nothing that has initialised HIP is
    ever re-executed."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; without it RCCL's buffer exchange
    # between the rank processes fails with `hipIpcGetMemHandle: invalid argument`. It is exported on the boxes already; kept
    # here so that a worker environment built from a scrubbed one still carries it.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def strong_scaling_legs(pkg, T, dtype, device, rank, world, dist, barrier, K, lib=None, globals_=(4096, 65536)):
    """Fixed GLOBAL batch split over the ranks (contiguous shards, no collective in the loop): K EM iterations each."""
    shard_range = importlib.import_module(PKG + ".dist").shard_range
    legs = []
    for Bg in globals_:
        lo, hi = shard_range(Bg, rank, world)
        eng = make_engine(pkg, max(hi - lo, 1), T, dtype, device, rank, lib=lib)
        for _ in range(2):
            eng.learn_msgs()
        barrier()
        t0 = time.perf_counter()
        eng.learn(K)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        legs.append({"global_batch": Bg, "batch_per_gpu": hi - lo, "ms_per_step": el / K * 1e3,
                     "value": Bg * T * K / el, "unit": "timestep-messages/s", "scaling": "strong",
                     "forward_family": eng.forward_family, "backward_family": eng.backward_family, "backward": eng.backward_schedule})
        del eng
    return legs


def _gbps(eng, B, T, ms, wbytes=8):
    d = eng.dims
    el = (d.e_post - eng.nu - eng.nu * (eng.nu + 1) // 2) + 2 * d.e_fwd + d.e_post
    return el * wbytes * B * T / (ms * 1e-3) / 1e9


def extra_config_legs(pkg, device, K=10):
    """The other BASELINE.json configs on one GPU, fp64, same timing discipline (synchronise, K steps, synchronise)."""
    from i2c.known_models import make_env_model

    out = {}

    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    rng = np.random.default_rng(7)

    # BASELINE.json configs[1]: pendulum_known_quad cubature i2c, nx=2 nu=1 T=200, batch B=1024 on one MI355X (the headline's
    # problem at a quarter of its batch: 16 lone wavefronts; the sweep is a fixed-length chain of T dependent cells)
    # (a sweep of 16 lone wavefronts is a chain of dependent instructions: its time is 1 / clock. The leg is short and light, so it
    # is preceded by 0.1 s of the same work on a throwaway engine -- measured without it: 0.39 ms right after the 131072-trajectory
    # legs, 0.59 ms after a 1 s pause, 0.33 ms in a loop of its own (tools/sweep_batch.py))
    warm = make_engine(pkg, 1024, 200, torch.float64, device, rank=2)
    warm.learn(300)
    del warm
    eng = make_engine(pkg, 1024, 200, torch.float64, device, rank=1)
    eng.learn(5)
    Kp = max(10 * K, 100)
    sync(); t0 = time.perf_counter(); eng.learn(Kp); sync()
    ms = (time.perf_counter() - t0) / Kp * 1e3
    gb = _gbps(eng, 1024, 200, ms)
    out["pendulum_T200_B1024"] = {"ms_per_step": ms, "value": 1024 * 200 / ms * 1e3, "unit": "timestep-messages/s", "steps": Kp,
                                  "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS, "backward": eng.backward_schedule,
                                  "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                                  "failed_trajectories": len(eng.failures())}
    del eng

    # config 3: double cartpole swing-up, T=300, B=4096 (scripts/experiments/double_cartpole_known_cq.py:23-39)
    m = make_env_model("DoubleCartpoleKnown")
    B, T = 4096, 300
    Q, R = 1e-3 * np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0]), 1e-3 * np.diag([0.1])
    x0 = np.asarray(m.x0, float).reshape(1, -1) + 1e-3 * rng.normal(size=(B, m.dim_x))
    eng = pkg.BatchedI2c(m, T, Q, R, Q, 0.05, 0.99, 1e-2 * rng.normal(size=(B, T, 1)), np.eye(1), x0=x0, device=device,
                         keep_zpost=False, keep_xm=False)
    eng.learn(5)
    Kd = max(2 * K, 20)  # (20 iterations = 34 ms: the 10-iteration figure read 4 % high right after the light B = 1024 leg -- clocks)
    sync(); t0 = time.perf_counter(); eng.learn(Kd); sync()
    ms = (time.perf_counter() - t0) / Kd * 1e3
    gb = _gbps(eng, B, T, ms)
    out["double_cartpole_T300_B4096"] = {"ms_per_step": ms, "steps": Kd, "value": B * T / ms * 1e3, "unit": "timestep-messages/s",
                                         "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS, "backward": eng.backward_schedule,
                                         "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                                         "failed_trajectories": len(eng.failures())}
    # its forward sweep alone (the quad kernel), for the issue-side roofline: event-timed
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    sync(); evs[0].record()
    for _ in range(5):
        eng.forward_sweep()
    evs[1].record(); sync()
    fwd_ms = evs[0].elapsed_time(evs[1]) / 5
    out["double_cartpole_T300_B4096"]["forward_sweep_ms"] = fwd_ms
    out["double_cartpole_T300_B4096"]["issue"] = issue_roofline("dcp_B4096", "k_quad_forward", T, 16, fwd_ms, B * T, useful=("dcp_B4096_group_and_lane", "k_forward"))
    del eng
    # config 3's "fp32 vs fp64 tolerance sweep", the speed side: the same problem with fp32-STORED messages (fp64 arithmetic; the
    # deviation from the fp64 run is bounded and asserted in tests/test_precision.py: median 1e-4, 99th percentile 3e-2 of the batch)
    eng = pkg.BatchedI2c(m, T, Q, R, Q, 0.05, 0.99, 1e-2 * np.random.default_rng(7).normal(size=(B, T, 1)), np.eye(1), x0=x0, device=device,
                         keep_zpost=False, keep_xm=False, storage_dtype=torch.float32)
    eng.learn(2)
    sync(); t0 = time.perf_counter(); eng.learn(K); sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    gb = _gbps(eng, B, T, ms, wbytes=4)
    out["double_cartpole_T300_B4096_fp32_storage"] = {"ms_per_step": ms, "value": B * T / ms * 1e3, "unit": "timestep-messages/s",
                                                      "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS,
                                                      "dtype": "f64 arithmetic, f32 storage (I2C_F64_F32S)", "backward": eng.backward_schedule,
                                                      "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                                                      "failed_trajectories": len(eng.failures())}
    del eng

    # the d >= 7 models at a batch that fills the chip (EM iteration = forward + backward + M-step, one i2c_learn call):
    # double cartpole T=300 and the 12-state quadrotor T=50 at B = 32768
    for tag, name, Tn, Bn in (("double_cartpole_T300_B32768", "DoubleCartpoleKnown", 300, 32768), ("quadrotor12_T50_B32768", "Quadrotor12", 50, 32768)):
        m = make_env_model(name)
        if name == "DoubleCartpoleKnown":
            Qn, Rn, Qfn, a0, tol, su = Q, R, Q, 0.05, 0.99, np.eye(1)
            mu_u = 1e-2 * rng.normal(size=(Bn, Tn, 1))
            x0 = np.asarray(m.x0, float).reshape(1, -1) + 1e-3 * rng.normal(size=(Bn, m.dim_x))
        else:
            Qn, Rn = np.diag([10.0] * 3 + [1.0] * 3 + [0.1] * 6), 1e-2 * np.eye(4)
            Qfn, a0, tol, su = Qn, 1.0, 0.5, 1e-2 * np.eye(4)
            mu_u = 0.25 * m.gravity + 1e-2 * rng.normal(size=(Bn, Tn, 4))
            x0 = 1e-2 * rng.normal(size=(Bn, 12))
        eng = pkg.BatchedI2c(m, Tn, Qn, Rn, Qfn, a0, tol, mu_u, su, x0=x0, device=device, keep_zpost=False, keep_xm=False)
        eng.learn(2)
        Ks = max(K // 2, 3)
        sync(); t0 = time.perf_counter(); eng.learn(Ks); sync()
        ms = (time.perf_counter() - t0) / Ks * 1e3
        gb = _gbps(eng, Bn, Tn, ms)
        out[tag] = {"ms_per_step": ms, "value": Bn * Tn / ms * 1e3, "unit": "timestep-messages/s", "algorithmic_GBps": gb,
                    "frac_of_hbm_peak": gb / HBM_PEAK_GBS, "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                    "backward": eng.backward_schedule, "failed_trajectories": len(eng.failures())}
        if name == "Quadrotor12":  # its sweeps alone, event-timed, and the issue side of the forward one (the quad kernel: issue-bound)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            sync(); evs[0].record()
            for _ in range(3):
                eng.forward_sweep()
            evs[1].record()
            for _ in range(3):
                eng.backward_sweep()
            evs[2].record(); sync()
            fwd_ms, bwd_ms = evs[0].elapsed_time(evs[1]) / 3, evs[1].elapsed_time(evs[2]) / 3
            C = eng.dims
            out[tag]["forward_sweep_ms"], out[tag]["backward_sweep_ms"] = fwd_ms, bwd_ms
            out[tag]["backward_sweep_algorithmic_GBps"] = (C.e_fwd + C.e_post) * 8 * Bn * Tn / bwd_ms / 1e6
            out[tag]["issue"] = issue_roofline("quad12_B32768_quad_vs_wave", "k_quad_forward", Tn, 16, fwd_ms, Bn * Tn)
            if out[tag]["issue"]:  # (d = 16 has no one-lane kernel to count the non-redundant flops with)
                out[tag]["issue"]["useful_flops_basis"] = "executed flops of this kernel (sixteen lanes per trajectory: an upper bound of the useful count)"
        del eng, mu_u, x0
        torch.cuda.empty_cache()

    # config 4: quadrotor MPC + cubature-KF state estimation at nx = 12, horizon 50, B = 8192 closed loops: one control step
    # = filter + n_iter x (forward, backward, prior update) + first action + horizon shift, ONE library call (i2c_mpc_step)
    # (B = 8192: the whole config on this GPU; B = 1024: one GPU's share of it when sharded over 8)
    m = make_env_model("Quadrotor12")
    for B in (8192, 1024):
        T, n_iter = 50, 2  # mpc_iter = 2 in the reference's script (scripts/mpc_state_est/mpc_quad.py:559)
        Q, R = np.diag([10.0] * 3 + [1.0] * 3 + [0.1] * 6), 1e-2 * np.eye(4)
        x0 = 1e-2 * rng.normal(size=(B, 12))
        mu_u = 0.25 * m.gravity + 1e-2 * rng.normal(size=(B, T, 4))
        eng = pkg.BatchedI2c(m, T, Q, R, Q / 10.0, 0.02, 1.0, mu_u, 1e-2 * np.eye(4), x0=x0, device=device, keep_zpost=False, keep_xm=False,
                             z_traj=np.broadcast_to(np.concatenate((m.zg_term.reshape(-1), 0.25 * m.gravity * np.ones(4))), (T, 16)))
        eng.tau = T - 1
        eng.enable_per_cell_alpha()
        sig_zeta = 1e-4 * np.eye(9)
        y = torch.as_tensor(np.ascontiguousarray(m.measure(x0).T), dtype=torch.float64, device=device)
        u = torch.as_tensor(np.ascontiguousarray(mu_u[:, 0, :].T), dtype=torch.float64, device=device)
        for _ in range(2):
            act, _ = eng.mpc_step(n_iter, y, u, sig_zeta)
        sync(); t0 = time.perf_counter()
        for _ in range(K):
            act, _ = eng.mpc_step(n_iter, y, u, sig_zeta)
        sync()
        ms = (time.perf_counter() - t0) / K * 1e3
        gb = _gbps(eng, B, T * n_iter, ms)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        sync(); evs[0].record()
        for _ in range(5):
            eng.forward_sweep()
        evs[1].record(); sync()
        fwd_ms12 = evs[0].elapsed_time(evs[1]) / 5
        out["quadrotor12_mpc_H50_B%d" % B] = {"ms_per_control_step": ms, "closed_loop_steps_per_s": B / ms * 1e3, "em_iters_per_step": n_iter,
                                              "value": B * T * n_iter / ms * 1e3, "unit": "timestep-messages/s",
                                              "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS,
                                              "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                                              "failed_trajectories": len(eng.failures()), "forward_sweep_ms": fwd_ms12}
        if B == 1024:  # (one wave per SIMD: the wave kernels; the issue side of their forward sweep from the committed SQ pass of this step)
            out["quadrotor12_mpc_H50_B1024"]["issue"] = issue_roofline("quad12_mpc_B1024", "k_wave<0", T, 64, fwd_ms12, B * T)
        del eng

    # config 4 on the reference's ACTUAL model class, the planar quadrotor (mpc_quad.py:219-383: nx = 6, nu = 2, identity observation),
    # same loop: B = 8192 and one GPU's share of it
    m = make_env_model("PlanarQuadrotor")
    for B in (8192, 1024):
        T, n_iter = 50, 2
        Q, R = np.diag([1e3, 1e3, 1e3, 1, 1, 1]) / 1e3, np.diag([1e-3, 1e-3])
        x0 = np.asarray(m.x0, float).reshape(1, -1) + 1e-2 * rng.normal(size=(B, 6))
        mu_u = 0.5 * m.gravity + 1e-2 * rng.normal(size=(B, T, 2))
        eng = pkg.BatchedI2c(m, T, Q, R, Q / 1e3, 1.0, 1.0, mu_u, 1e-2 * np.eye(2), x0=x0, device=device, keep_zpost=False, keep_xm=False,
                             z_traj=np.broadcast_to(np.concatenate((np.asarray(m.x0, float).reshape(-1), 0.5 * m.gravity * np.ones(2))), (T, 8)))
        eng.tau = T - 1
        eng.enable_per_cell_alpha()
        sig_zeta = 1e-4 * np.eye(8)
        y = torch.as_tensor(np.ascontiguousarray(m.measure(x0).T), dtype=torch.float64, device=device)
        u = torch.as_tensor(np.ascontiguousarray(mu_u[:, 0, :].T), dtype=torch.float64, device=device)
        for _ in range(2):
            act, _ = eng.mpc_step(n_iter, y, u, sig_zeta)
        sync(); t0 = time.perf_counter()
        for _ in range(K):
            act, _ = eng.mpc_step(n_iter, y, u, sig_zeta)
        sync()
        ms = (time.perf_counter() - t0) / K * 1e3
        gb = _gbps(eng, B, T * n_iter, ms)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        sync(); evs[0].record()
        for _ in range(5):
            eng.forward_sweep()
        evs[1].record(); sync()
        fwd_msp = evs[0].elapsed_time(evs[1]) / 5
        out["planar_quadrotor_mpc_H50_B%d" % B] = {"ms_per_control_step": ms, "closed_loop_steps_per_s": B / ms * 1e3, "em_iters_per_step": n_iter,
                                                   "value": B * T * n_iter / ms * 1e3, "unit": "timestep-messages/s",
                                                   "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS,
                                                   "forward_family": eng.forward_family, "backward_family": eng.backward_family,
                                                   "failed_trajectories": len(eng.failures()), "forward_sweep_ms": fwd_msp}
        if B == 1024:
            out["planar_quadrotor_mpc_H50_B1024"]["issue"] = issue_roofline("planar_mpc_B1024", "k_quad_forward", T, 16, fwd_msp, B * T)
        del eng

    # the other inference rules of the reference on the headline shape (pendulum T=200, B=4096; exp_types.py:22-68): one EM iteration
    # with Linearize() (i2c.py:244-348, 449-542) and with GaussHermiteQuadrature(3) (27 / 8 points per transform)
    mp = make_env_model("PendulumKnown")
    for tag, kw in (("pendulum_T200_B4096_linearize", dict(inference="linearize")), ("pendulum_T200_B4096_gauss_hermite3", dict(inference="gauss_hermite", gh_degree=3))):
        Bp, Tp = 4096, 200
        x0p, mu_up = synthetic_pendulum_inputs(Bp, Tp, 3)
        eng = pkg.BatchedI2c(mp, Tp, np.diag([1.0, 100.0, 1.0]), np.diag([2.0]), np.diag([1.0, 100.0, 1.0]), 100.0, 0.0, mu_up, 2.0 * np.eye(1), x0=x0p,
                             device=device, keep_zpost=False, keep_xm=False, **kw)
        eng.learn(2)
        sync(); t0 = time.perf_counter(); eng.learn(K); sync()
        ms = (time.perf_counter() - t0) / K * 1e3
        gb = _gbps(eng, Bp, Tp, ms)
        out[tag] = {"ms_per_step": ms, "value": Bp * Tp / ms * 1e3, "unit": "timestep-messages/s", "algorithmic_GBps": gb,
                    "frac_of_hbm_peak": gb / HBM_PEAK_GBS, "backward": eng.backward_schedule, "failed_trajectories": len(eng.failures())}
        del eng

    # config 5: nonlinear covariance control (pendulum, action-only cost, annealed terminal prior, closed-loop propagation
    # and KL every iteration; scripts/experiments/pendulum_known_act_reg_quad.py:22-33), T=100, B=8192 = one GPU's share of 65536
    m = make_env_model("PendulumKnownActReg")
    B, T = 8192, 100
    x0 = np.array([np.pi, 0.0]) + 1e-2 * rng.normal(size=(B, 2))
    eng = pkg.BatchedI2c(m, T, None, np.diag([1.0]), None, 300.0, 1.0, np.zeros((B, T, 1)), 0.5 * np.eye(1), np.array([0.0, 0.0]),
                         np.diag([1e-3, 1e-3]), x0=x0, device=device, keep_zpost=False)
    eng.use_expert_controller = False
    eng._propagate = True
    eng.propagate()
    eng.learn(3)
    Kc = max(K, 20)
    sync(); t0 = time.perf_counter()
    eng.learn(Kc)  # (round 5: the propagation of iteration k runs on a second stream next to the forward sweep of iteration k + 1)
    sync()
    ms = (time.perf_counter() - t0) / Kc * 1e3
    gb = (_gbps(eng, B, T, ms) + eng.dims.e_prop * 8 * B * T / (ms * 1e-3) / 1e9)  # + the propagation rows written each iteration
    out["covariance_control_T100_B8192"] = {"ms_per_step": ms, "value": B * T / ms * 1e3, "unit": "timestep-messages/s",
                                            "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBS,
                                            "includes": "forward, backward, closed-loop propagation (overlapped with the next forward sweep on a second stream), KL, M-step",
                                            "failed_trajectories": len(eng.failures())}
    return out


def reference_shape_legs(pkg, device, lib=None, scale=1.0):
    """The shapes the REFERENCE itself runs -- one trajectory (B = 1) -- driven the way its scripts drive them: through the drop-in
    facade, `I2cGraph.learn_msgs()` in a Python loop (scripts/i2c_run.py:89-94) and `PartiallyObservedMpcPolicy.__call__`
    (i2c/policy/mpc.py:156-182), each beside the reference's own CPU figure of BASELINE.md section 2 (one core of the survey
    container; a stated baseline, not measured on this box). `kernel_ms` = the same iterations enqueued by ONE library call on a
    plain engine (BatchedI2c.learn: no Python between the sweeps); `host_ms` = what the facade's per-iteration Python, ctypes calls,
    failure check (one synchronisation per iteration: the reference raises inside the iteration) and bookkeeping add to it."""
    from i2c.exp_types import CubatureQuadrature
    from i2c.i2c import I2cGraph
    from i2c.known_models import make_env_model
    from i2c.policy.mpc import PartiallyObservedMpcPolicy

    device = torch.device(device)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)  # noqa: E731
    out = {}

    def em_leg(tag, model, T, Q, R, Qf, alpha, tol, sig_u, mu_u, ref_its, n=60):
        n = max(int(n * scale), 2)
        g = I2cGraph(make_env_model(model), T, Q, R, Qf, alpha, tol, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0), device=device, lib=lib)
        for _ in range(max(int(100 * scale), 5)):  # (warm: a lone wave's sweep is a chain of dependent instructions, its time is 1 / clock)
            g.learn_msgs()
        g.reset_metrics()
        sync(); t0 = time.perf_counter()
        for _ in range(n):
            g.learn_msgs()
        sync()
        facade_ms = (time.perf_counter() - t0) / n * 1e3
        n_lists = len(g.costs_m)  # (the lists the runner prints: read once, after the loop, as scripts/i2c_run.py:108-114 does)
        eng = pkg.BatchedI2c(make_env_model(model), T, Q, R, Qf, alpha, tol, mu_u[None], sig_u, device=device, keep_zpost=False, keep_xm=False, lib=lib)
        eng.learn(5)
        sync(); t0 = time.perf_counter(); eng.learn(n); sync()
        kernel_ms = (time.perf_counter() - t0) / n * 1e3
        out[tag] = {"facade_ms_per_em_iter": facade_ms, "em_iters_per_s": 1e3 / facade_ms, "kernel_ms": kernel_ms, "host_ms": facade_ms - kernel_ms,
                    "facade_over_engine_learn": facade_ms / kernel_ms, "reference_em_iters_per_s_1core": ref_its,
                    "vs_reference_1core": (1e3 / facade_ms) / ref_its, "iterations": n, "history_entries": n_lists,
                    "forward_family": g.engine.forward_family, "backward_family": g.engine.backward_family, "backward": g.engine.backward_schedule,
                    "failed_trajectories": len(g.engine.failures())}

    rs = np.random.RandomState(0)
    Qp = np.diag([1.0, 100.0, 1.0])
    em_leg("pendulum_T100_B1", "PendulumKnown", 100, Qp, np.diag([2.0]), Qp, 100.0, 0.0, 2.0 * np.eye(1), 1e-2 * rs.randn(100, 1), 13.6)
    em_leg("pendulum_T200_B1", "PendulumKnown", 200, Qp, np.diag([2.0]), Qp, 100.0, 0.0, 2.0 * np.eye(1), 1e-2 * rs.randn(200, 1), 6.35)
    Qd = 1e-3 * np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0])
    em_leg("double_cartpole_T300_B1", "DoubleCartpoleKnown", 300, Qd, 1e-3 * np.diag([0.1]), Qd, 0.05, 0.99, np.eye(1), 1e-2 * rs.randn(300, 1), 3.75, n=30)

    # MPC control step, H = 10, n_iter = 2, one closed loop, feed-forward and feedback (6 / 12 ms on the stand-in model)
    for tag, ff, ref_ms in (("mpc_pendulum_H10_B1_feedforward", True, 6.0), ("mpc_pendulum_H10_B1_feedback", False, 12.0)):
        model = make_env_model("PendulumKnown")
        model.sig_zeta = 1e-4 * np.eye(model.dim_y if hasattr(model, "dim_y") else 3)
        H = 10
        i2c = I2cGraph(model, H, Qp, np.diag([2.0]), Qp, 10.0, 1.0, np.zeros((H, 1)), 2.0 * np.eye(1), None, None, CubatureQuadrature(1, 0, 0), device=device, lib=lib)
        i2c._propagate = True
        steps = max(int(60 * scale), 14)
        z_traj = np.tile(np.asarray(model.zg, float).reshape(1, -1), (steps + H + 10, 1))
        pol = PartiallyObservedMpcPolicy(i2c, 2, 2.0 * np.eye(1), z_traj)
        pol.set_control(feedforward=ff)
        i2c.calibrate_alpha()
        pol.optimize(8, model.x0, model.sig_x0)
        i2c.calibrate_alpha()
        x = np.asarray(model.x0, float).reshape(1, -1)
        u = np.zeros((1, 1))
        ts = []
        for t in range(steps):
            y = model.measure(x).reshape(-1, 1)
            sync(); t0 = time.perf_counter()
            u = np.clip(pol(t, y, u.reshape(-1, 1)), -2.0, 2.0)
            sync()
            ts.append(time.perf_counter() - t0)
            x = model.dynamics(np.concatenate((x, u.reshape(1, -1)), axis=1)).reshape(1, -1)
        ms = float(np.median(ts[10:])) * 1e3
        out[tag] = {"ms_per_control_step": ms, "reference_ms_per_control_step_1core": ref_ms, "vs_reference_1core": ref_ms / ms, "steps": steps - 10,
                    "em_iters_per_step": 2, "horizon": H, "history_recorded": bool(pol.record_history), "failed_trajectories": len(i2c.engine.failures())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories PER GPU")
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--dtype", default="f64", choices=["f64"], help="the path computes in fp64 (the reference's arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the extra B=131072 leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs of the other BASELINE configs and strong scaling")
    ap.add_argument("--no-rccl", action="store_true", help="N = 1 only: do not create the single-rank RCCL group")
    ap.add_argument("--backward", default="auto", choices=["auto", "two_pass", "fused", "chunked"])
    ap.add_argument("--group-lanes", type=int, default=0, help="run the group kernels (lanes per trajectory) in the headline leg")
    ap.add_argument("--strong-global", type=int, nargs="*", default=None,
                    help="global batch sizes of the strong-scaling legs (default: 4096 65536)")
    ap.add_argument("--test-hostsim", action="store_true",
                    help="TESTS ONLY: gloo + the host simulation of the kernels on CPU, to exercise the launcher / N > 1 "
                         "plumbing without a GPU; its numbers mean nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:  # not under torchrun yet: become the launcher (before any GPU call)
        sys.exit(launch_workers(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    lib = None
    if args.test_hostsim:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        lib = importlib.import_module("hostsim").load()
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    rccl_error = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:  # launched by torch.distributed.run (also with N = 1)
        import torch.distributed as dist

        if args.test_hostsim:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # RCCL on ROCm
    elif args.gpus == 1 and not args.test_hostsim and not args.no_rccl:
        # A plain `python bench.py` (how the driver runs N = 1): a single-rank RCCL group in this process, so that the job's one
        # collective -- the all-gather of the final controllers -- runs through the same code path as N > 1 and the line
        # carries its time. Nothing is re-executed; the group is created before any kernel of the benchmark runs.
        import socket

        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for attempt in range(3):  # (the port is picked by bind-then-close: another job may take it in between -- try again)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            try:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
                rccl_error = None
                break
            except Exception as e:  # the measurement must not die with the collective library: say so in the line instead
                rccl_error = f"{type(e).__name__}: {e}"
        if rccl_error is not None:
            dist = None

    pkg = importlib.import_module(PKG)
    dtype = torch.float64
    B, T = args.batch, args.horizon
    eng = make_engine(pkg, B, T, dtype, device, rank, args.backward, lib=lib, group_lanes=args.group_lanes)

    def barrier():
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        eng.learn_msgs()

    # ---- per-kernel timing: K EM iterations stepped from Python with HIP events around each sweep -----
    K = args.steps
    if device.type == "cuda":
        stepwise_s, (fwd_ms, bwd_ms, mst_ms) = timed_iterations(eng, K, barrier)
    else:
        stepwise_s, (fwd_ms, bwd_ms, mst_ms) = 0.0, (float("nan"),) * 3

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
    # ---- the same loop over max(K, 200) iterations (round-4 review, weak #12: the contract's K = 20 steps are a 7 ms region, in which
    # one stray 50 us hiccup is 0.7 %): reported NEXT TO the K-step figure, which stays the line's `value` -----------------------------
    n_long = max(K, 200)
    barrier()
    t0 = time.perf_counter()
    eng.learn(n_long)
    barrier()
    elapsed_long = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed_long], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_long = float(tmax.item())
    n_fail = len(eng.failures())

    # ---- the one collective of the job: all-gather of the final controllers -------
    allgather_ms = None
    if dist is not None:
        gather_policy = importlib.import_module(PKG + ".dist").gather_policy

        gather_policy(eng, total=B * world)  # first call pays RCCL's lazy channel setup; time the steady-state exchange
        barrier()
        t1 = time.perf_counter()
        gathered = gather_policy(eng, total=B * world)  # (weak scaling: every rank holds B trajectories)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        allgather_ms = (time.perf_counter() - t1) * 1e3
        assert gathered["K"].shape[0] == B * world

    # (as a tensor of small integers through the same collective primitive as the job's all-gather: no pickling, no object collectives)
    FAM, SCH = ["lane", "group", "wave", "quad"], ["two_pass", "fused", "chunked"]
    mine = torch.tensor([rank, B, FAM.index(eng.forward_family), FAM.index(eng.backward_family), SCH.index(eng.backward_schedule)],
                        dtype=torch.int64, device=device)
    rows = mine.reshape(1, -1)
    if dist is not None and world > 1:
        rows = torch.empty(world, mine.numel(), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(rows, mine.reshape(1, -1).contiguous())
    families = [{"rank": int(r[0]), "batch": int(r[1]), "forward_family": FAM[int(r[2])], "backward_family": FAM[int(r[3])], "backward": SCH[int(r[4])]}
                for r in rows.cpu().tolist()]
    strong = None
    if not args.no_extra:  # every rank takes part (collective timing)
        strong = strong_scaling_legs(pkg, T, dtype, device, rank, world, dist, barrier, max(K // 5, 2), lib=lib,
                                     globals_=tuple(args.strong_global) if args.strong_global else ((4096, 65536) if not args.test_hostsim else (8,)))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    cells = B * T * world
    wbytes = 8
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
        "long_run": {"steps": n_long, "ms_per_step": elapsed_long / n_long * 1e3, "value": cells * n_long / elapsed_long,
                     "note": "the same timed loop over max(steps, 200) EM iterations: the noise floor of the K-step figure"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic" if not args.test_hostsim else "synthetic (TEST MODE: host simulation on CPU, numbers meaningless)",
        "config": {
            "workload": f"pendulum_known_quad cubature i2c (nx=2, nu=1, nz=4), T={T}, B={B} trajectories per GPU, "
                        f"global batch {B * world}, CubatureQuadrature(1,0,0), alpha0=100, tol=0",
            "batch_per_gpu": B,
            "horizon": T,
            "parallelism": f"batch-sharded x{world}, no collective in the EM loop",
            "kernels": "group, %d lanes per trajectory" % args.group_lanes if args.group_lanes else "one lane per trajectory",
        },
        "rccl_world_size": (dist.get_world_size() if dist is not None else None),
        "rccl_backend": (dist.get_backend() if dist is not None else None),
        "rccl_error": rccl_error,
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
                            "(FETCH_SIZE x2, gfx950 correction; the newest profiles/rN_B4096_pmc_traffic.json); algorithmic = "
                            + str(fwd_bytes),
            "algorithmic_bytes_per_cell": {k: v * wbytes for k, v in el.items()},
            "whole_iteration_GBps": el["total"] * wbytes * B * T / (elapsed / K) / 1e9,
            "issue": issue_roofline("pendulum_B4096", "k_forward", T, 1, fwd_ms, B * T) if (B, T) == (4096, 200) else None,
            "note": "instruction-issue-bound at B=4096 per GPU: 64 lone wavefronts (1024 SIMDs), each issuing 88 % of its cycles "
                    "(profiles/r4_pendulum_B4096_sq_counters.json, r3_pendulum_B4096_chunked_sq_summary.txt; `issue`: what the fp64 "
                    "pipe of an occupied SIMD is doing); spreading a trajectory over lanes is SLOWER for this 3-dimensional model "
                    "(group kernels: profiles/r2_pendulum_lane_vs_group_sq_counters.json; quad kernel, 4 x 4 blocks on the matrix "
                    "instruction: profiles/r4_quad_forward_timings_v3.txt); HBM-bound from B ~ 32768 (saturated_batch); DESIGN.md 6",
        },
        "final_allgather_ms": allgather_ms,
        "strong_scaling": strong,
        # what every rank ran (the default family and the backward schedule depend on the shard size: DESIGN.md section 8)
        "families_per_rank": families,
    }
    if device.type == "cuda":
        if not args.no_saturated and world == 1:
            out["saturated_batch"] = saturated_leg(pkg, T, dtype, device, el, wbytes)
            mixed = saturated_leg(pkg, T, dtype, device, el, 4, storage_dtype=torch.float32)
            mixed.pop("device_memcpy_GBps")
            mixed["precision"] = ("fp64 arithmetic on fp32-stored messages (I2C_F64_F32S): NOT the parity-grade path; deviation of the "
                                  "posterior mean from the fp64 run over 12 EM iterations at T=200, B=4096: <= 2e-5 after one iteration; "
                                  "median over the batch <= 1e-4, 99th percentile <= 3e-2 (asserted, tests/test_precision.py); the few "
                                  "trajectories at the swing-up bifurcation deviate O(1) under ANY perturbation")
            out["saturated_batch_fp32_storage"] = mixed
        if not args.no_extra and world == 1:
            out["extra"] = extra_config_legs(pkg, device)
            out["extra"]["reference_shapes"] = reference_shape_legs(pkg, device)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(T)
    # the JSON line is the LAST thing on stdout: tear the process group down first and flush the C runtime's buffer (RCCL prints a
    # version banner through it, which otherwise lands after Python's output when stdout is a pipe)
    if dist is not None:
        dist.destroy_process_group()
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
