"""`python bench.py --gpus N` must start its own N workers (the driver's plain form): exercised here with N = 2 on CPU --
gloo + the host simulation of the kernels (bench.py --test-hostsim, a test-only switch) -- so that the launcher, the
rendezvous on 127.0.0.1, the max-over-ranks timing, the final all-gather and the strong-scaling legs are covered without
a GPU. The numbers of this mode mean nothing; the JSON contract is what is checked."""
import json
import os
import subprocess
import sys

import hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    hostsim.build()  # once, in the parent, so that the two workers do not race to build it
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on rank 0"
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--test-hostsim", "--batch", "6", "--horizon", "12", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["scaling"] == "weak"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["config"]["batch_per_gpu"] == 6 and "global batch 12" in out["config"]["workload"]
    assert out["value"] > 0 and out["final_allgather_ms"] is not None
    assert out["strong_scaling"][0]["global_batch"] == 8 and out["strong_scaling"][0]["batch_per_gpu"] == 4
    # every rank reports what it ran (the default family / schedule depend on the shard size: a SCALE record must show them)
    fam = out["families_per_rank"]
    assert [f["rank"] for f in fam] == [0, 1] and all(f["batch"] == 6 and f["forward_family"] == "lane" and f["backward"] for f in fam)
    assert out["strong_scaling"][0]["forward_family"] == "lane"
    assert "TEST MODE" in out["data"]


def test_bench_self_launches_eight_ranks_ragged():
    """The N = 8 launcher path of the driver's scaling run (no 8-GPU lease has been available to measure it): eight gloo ranks on
    the host simulation, a global strong-scaling batch that does not divide by 8 (4099 = 3 x 513 + 5 x 512)."""
    out = _run(["--gpus", "8", "--test-hostsim", "--batch", "3", "--horizon", "8", "--steps", "2", "--warmup", "1", "--strong-global", "4099"])
    assert out["n_gpus"] == 8 and out["rccl_world_size"] == 8 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 3 and "global batch 24" in out["config"]["workload"]
    leg = out["strong_scaling"][0]
    assert leg["global_batch"] == 4099 and leg["batch_per_gpu"] == 513 and leg["scaling"] == "strong" and leg["value"] > 0
    assert out["final_allgather_ms"] is not None and out["failed_trajectories"] == 0
    assert len(out["families_per_rank"]) == 8 and {f["rank"] for f in out["families_per_rank"]} == set(range(8))


def test_bench_single_process_contract():
    out = _run(["--test-hostsim", "--batch", "5", "--horizon", "10", "--steps", "2", "--warmup", "1", "--no-extra"])
    assert out["n_gpus"] == 1 and out["final_allgather_ms"] is None and out["strong_scaling"] is None
    assert out["rccl_world_size"] is None  # no process group in the host-simulation mode
    for key in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "vs_baseline"):
        assert key in out
    assert out["dtype"] == "f64" and out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0


# ---- on the GPU box: the RCCL path of the N = 1 line (round-2 review: it had never been executed on hardware) ----------
import socket  # noqa: E402

import pytest  # noqa: E402

_FAST = ["--steps", "2", "--warmup", "1", "--no-extra", "--no-saturated", "--no-cpu-baseline", "--batch", "256", "--horizon", "40"]


def _child(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)  # a fresh child: nothing is re-executed
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_plain_single_gpu_line_carries_the_rccl_all_gather():
    """`python bench.py` as the driver runs it at N = 1: a single-rank RCCL group in-process, the all-gather timed."""
    out = _child([sys.executable, os.path.join(ROOT, "bench.py")] + _FAST)
    assert out["rccl_error"] is None, out["rccl_error"]
    assert out["rccl_world_size"] == 1 and out["rccl_backend"] == "nccl"
    assert out["final_allgather_ms"] is not None and out["final_allgather_ms"] > 0
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["failed_trajectories"] == 0


@pytest.mark.gpu
def test_bench_under_torchrun_with_one_rank():
    """The driver's N > 1 form with --nproc-per-node=1: rendezvous on 127.0.0.1, init_process_group("nccl"), all-gather."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = _child([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                  "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + _FAST)
    assert out["rccl_world_size"] == 1 and out["rccl_backend"] == "nccl" and out["final_allgather_ms"] > 0
