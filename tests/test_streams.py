"""The ABI says "calls on different streams are independent" (include/i2c_hip.h). This test backs the
sentence (round-4 review, weak #11): two engines of different kernel families -- the pendulum on the lane kernels (chunked
backward schedule with its per-engine `work` buffer) and the 12-state quadrotor on the quad kernels (per-workgroup LDS state) --
are stepped ALTERNATELY on two non-default HIP streams with no synchronisation in between, so that their kernels overlap on the
device; every result must be bit-identical to the same engines run one after the other on the default stream."""
import importlib

import numpy as np
import pytest
import torch

pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import make_env_model  # noqa: E402


def make_engines(device):
    rng = np.random.default_rng(11)
    B, T = 2048, 60
    x0 = np.array([np.pi, 0.0]) + 1e-2 * rng.normal(size=(B, 2))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])
    pend = pkg.BatchedI2c(make_env_model("PendulumKnown"), T, Q, R, Q, 100.0, 0.0, mu_u, 2.0 * np.eye(1), x0=x0, device=device)
    q = make_env_model("Quadrotor12")
    Bq, Tq = 1024, 20
    x0q = 1e-2 * rng.normal(size=(Bq, 12))
    mu_uq = 0.25 * q.gravity + 1e-2 * rng.normal(size=(Bq, Tq, 4))
    Qq, Rq = np.diag([10.0] * 3 + [1.0] * 3 + [0.1] * 6), 1e-2 * np.eye(4)
    quad = pkg.BatchedI2c(q, Tq, Qq, Rq, Qq, 1.0, 0.5, mu_uq, 1e-2 * np.eye(4), x0=x0q, device=device,
                          group_lanes=pkg._native.LANES_QUAD)
    # a third engine of the SAME model and family as the first: the same kernels on both streams at once
    pend2 = pkg.BatchedI2c(make_env_model("PendulumKnown"), T, Q, R, Q, 50.0, 0.0, -mu_u, 2.0 * np.eye(1), x0=x0[::-1].copy(), device=device)
    return pend, quad, pend2


def snapshot(e):
    K, k, sigK = e.local_linear_policy()
    mu, sig = e.marginal_state_action()
    return [t.detach().cpu().numpy().copy() for t in (K, k, sigK, mu, sig, e.alpha)]


@pytest.mark.gpu
def test_two_streams_interleaved_match_serial_runs():
    dev = "cuda:0"
    n_iters = 4
    serial = []
    for e in make_engines(dev):
        for _ in range(n_iters):
            e.learn_msgs()
        torch.cuda.synchronize()
        assert e.failures() == []
        serial.append(snapshot(e))
    assert make_engines(dev)[1].forward_family == "quad"

    pend, quad, pend2 = make_engines(dev)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    assert s1.cuda_stream != s2.cuda_stream and s1.cuda_stream != torch.cuda.default_stream(dev).cuda_stream
    for _ in range(n_iters):  # alternate, never synchronising: the sweeps of one engine are in flight while the other's are enqueued
        with torch.cuda.stream(s1):
            pend.forward_sweep()
        with torch.cuda.stream(s2):
            quad.forward_sweep()
        with torch.cuda.stream(s1):
            pend.backward_sweep()
        with torch.cuda.stream(s2):
            quad.backward_sweep()
            quad.maximize()
        with torch.cuda.stream(s1):
            pend.maximize()
    # the same model and family on both streams at once: pend2 follows quad on s2 while pend keeps s1 busy with four more iterations
    for _ in range(n_iters):
        with torch.cuda.stream(s2):
            pend2.learn_msgs()
        with torch.cuda.stream(s1):
            pend.learn_msgs()
    s1.synchronize()
    s2.synchronize()
    for e in (pend, quad, pend2):
        assert e.failures() == []
    for got, want, name in ((snapshot(quad), serial[1], "quad12 (quad kernels)"), (snapshot(pend2), serial[2], "pendulum #2 (lane kernels)")):
        for a, b in zip(got, want):
            assert np.array_equal(a, b), f"{name}: a result changed when its sweeps overlapped another engine's"
    # pend ran 2 * n_iters iterations in the end: compare with a serial run of the same length
    ref = make_engines(dev)[0]
    for _ in range(2 * n_iters):
        ref.learn_msgs()
    torch.cuda.synchronize()
    for a, b in zip(snapshot(pend), snapshot(ref)):
        assert np.array_equal(a, b), "pendulum (lane kernels): a result changed when its sweeps overlapped another engine's"
