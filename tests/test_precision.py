"""BASELINE config 2's "fp32 vs fp64 tolerance sweep", as asserted bounds per EM iteration.

Three precisions exist (include/i2c_hip.h, I2cProblem.dtype):
  * I2C_F64        fp64 arithmetic + storage: the reference's arithmetic, pinned to it by the parity tests;
  * I2C_F64_F32S   fp64 ARITHMETIC on fp32-STORED per-cell messages (BatchedI2c(storage_dtype=torch.float32)): an opt-in
                   mode that halves the HBM bytes of the sweeps; its deviation from fp64 is BOUNDED (asserted below) but
                   outside the 1e-5 parity bar, because the swing-up EM amplifies the 6e-8 rounding of the stored messages;
  * I2C_F32        fp32 arithmetic: NOT parity-grade -- the curvature terms of the sigma-point transform fall below fp32
                   resolution and the run drifts O(1) away within a few EM iterations while status stays 0. The engine
                   refuses it unless allow_inexact=True is passed.
The CPU tests run the host simulation at small batches; the -m gpu tests run the full shapes of the config
(pendulum T=200 and double cartpole T=300, B=4096)."""
import importlib

import numpy as np
import pytest
import torch

import hostsim
import parity
from golden_util import load_case

pkg = importlib.import_module("input-inference-for-control_amd")


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max())


def _rel_per_traj(a, b):
    """(B,) deviation of every trajectory: max over its entries of |a - b|, relative to the batch-wide max |b|."""
    a, b = a.double(), b.double()
    return (a - b).abs().flatten(1).max(dim=1).values / b.abs().max()


def _sweep(name, lib, device, B, iters, **kw):
    g = load_case(name)
    x0, mu_u = parity.batched_inputs(g, B)
    e64 = parity.engine_from_case(g, lib, device, x0=x0, mu_u=mu_u)
    alt = parity.engine_from_case(g, lib, device, x0=x0, mu_u=mu_u, **kw)
    rows = []
    for it in range(1, iters + 1):
        e64.learn_msgs()
        alt.learn_msgs()
        m64, s64 = e64.marginal_state_action()
        ma, sa = alt.marginal_state_action()
        K64, k64, _ = e64.local_linear_policy()
        Ka, ka, _ = alt.local_linear_policy()
        dm = _rel_per_traj(ma, m64)
        dc = (alt.costs_m[-1].double() - e64.costs_m[-1]).abs() / e64.costs_m[-1].abs()
        rows.append(dict(it=it, mean=float(dm.max()), mean_median=float(dm.median()), mean_p99=float(dm.quantile(0.99)),
                         cov=_rel(sa, s64), K=_rel(Ka, K64), k=_rel(ka, k64), alpha=_rel(alt.alpha, e64.alpha),
                         cost=float(dc.max()), cost_median=float(dc.median())))
    assert e64.failures() == [] and alt.failures() == []
    return rows, alt


# Bounds of the MIXED mode at every EM iteration: (median over the batch of the posterior-mean deviation, its 99th
# percentile, median cost deviation). The swing-up EM amplifies the 6e-8 rounding of the stored messages, and a handful
# of the 4096 perturbed problems sit where the swing-up direction is decided: THEIR deviation is O(1) under any
# perturbation (measured on MI355X: worst trajectory 0.35 pendulum / 0.5 double cartpole, fp32 arithmetic 0.5 / 1.1),
# so the worst case is reported, not bounded; the distribution is what is asserted
# (measured: median 5e-6 / 1e-4, 99th percentile 3e-3 / 5e-3, median cost deviation 6e-7 / 9e-6).
MIXED_BOUNDS = {"em_pendulum_T200": (1e-4, 3e-2, 1e-5), "em_dcp_T300_run20": (1e-3, 3e-2, 1e-4), "em_dcp_T60": (1e-3, 3e-2, 1e-4),
                "em_quad12_T20": (1e-4, 1e-3, 1e-5)}  # 12-state quadrotor, wave kernels: 10 KB of messages per cell, the most HBM-hungry


def _check_mixed(name, lib, device, B, iters, **kw):
    rows, eng = _sweep(name, lib, device, B, iters, storage_dtype=torch.float32, **kw)
    assert eng.mixed and eng.post.dtype == torch.float32 and eng.fwd.dtype == torch.float32 and eng.alpha.dtype == torch.float64
    b_med, b_p99, b_cost = MIXED_BOUNDS[name]
    for r in rows:
        assert r["mean_median"] <= b_med and r["mean_p99"] <= b_p99 and r["cost_median"] <= b_cost, r
    assert rows[0]["mean"] <= 2e-5 and rows[0]["cost"] <= 2e-6, rows[0]  # one iteration: storage rounding only
    return rows


def test_fp32_arithmetic_has_to_be_asked_for():
    g = load_case("em_pendulum_T200")
    with pytest.raises(ValueError, match="allow_inexact"):
        parity.engine_from_case(g, hostsim.load(), "cpu", dtype=torch.float32)
    with pytest.raises(ValueError, match="one-lane"):
        parity.engine_from_case(g, hostsim.load(), "cpu", storage_dtype=torch.float32, group_lanes=True)


def test_mixed_precision_bounds_cpu():
    _check_mixed("em_pendulum_T200", hostsim.load(), "cpu", 6, 6)


def test_mixed_precision_wave_kernels_cpu():
    """fp32-stored messages for the 12-state quadrotor: the wave kernels are templated on the storage type (round-2 review: the
    mixed mode did not reach the model that moves the most bytes). The group kernels stay fp64-only."""
    rows = _check_mixed("em_quad12_T20", hostsim.load(), "cpu", 4, 4)
    assert rows[-1]["mean"] < 1e-4, rows[-1]
    with pytest.raises(ValueError, match="wave"):
        parity.engine_from_case(load_case("em_quad12_T20"), hostsim.load(), "cpu", storage_dtype=torch.float32, group_lanes=16)


def test_mixed_precision_quad_kernels_cpu():
    """fp32-stored messages through the quad kernels (round 4, late): the forward sweep of the d <= 8 models (the default of the
    double cartpole up to 8192 trajectories: without it the opt-in mode ran the slower lane forward sweep there) and both sweeps of
    the 12-state quadrotor (cell blocks staged through LDS as pairs of floats)."""
    rows, eng = _sweep("em_dcp_T60", hostsim.load(), "cpu", 5, 3, storage_dtype=torch.float32)
    assert eng.forward_family == "quad" and eng.fwd.dtype == torch.float32
    _check_mixed("em_dcp_T60", hostsim.load(), "cpu", 5, 3)
    rows, eng = _sweep("em_quad12_T20", hostsim.load(), "cpu", 5, 3, storage_dtype=torch.float32, group_lanes=parity.pkg._native.LANES_QUAD)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad") and eng.post.dtype == torch.float32
    b_med, b_p99, b_cost = MIXED_BOUNDS["em_quad12_T20"]
    for r in rows:
        assert r["mean_median"] <= b_med and r["mean_p99"] <= b_p99 and r["cost_median"] <= b_cost, r


def test_mixed_precision_schedules_agree_cpu():
    """fp32 storage goes through the same three backward schedules; with fp32-stored messages they agree to storage
    rounding (not bit for bit: the chunked schedule composes the recursion in fp64 across cells)."""
    g = load_case("em_pendulum_T200")
    outs = []
    for mode in ("fused", "two_pass", "chunked"):
        e = parity.engine_from_case(g, hostsim.load(), "cpu", storage_dtype=torch.float32, backward_mode=mode)
        e.learn_msgs()
        outs.append(e.marginal_state_action()[0].double())
    assert _rel(outs[1], outs[0]) < 1e-5 and _rel(outs[2], outs[0]) < 1e-5


def test_mixed_mode_refuses_what_it_does_not_cover_cpu():
    g = load_case("em_pendulum_T50_propagate")
    e = parity.engine_from_case(g, hostsim.load(), "cpu", storage_dtype=torch.float32)
    with pytest.raises(RuntimeError, match="-2"):
        e.propagate()


@pytest.mark.gpu
@pytest.mark.parametrize("name,iters", [("em_pendulum_T200", 12), ("em_dcp_T300_run20", 8)])
def test_mixed_precision_bounds_full_config_gpu(name, iters):
    rows = _check_mixed(name, None, "cuda", 4096, iters)
    for r in rows:
        print(name, "it %2d: mean deviation median %.1e p99 %.1e max %.1e | cost median %.1e max %.1e" %
              (r["it"], r["mean_median"], r["mean_p99"], r["mean"], r["cost_median"], r["cost"]))


@pytest.mark.gpu
def test_mixed_precision_quad_kernels_gpu():
    """The 12-state quadrotor at a batch whose default is the quad family for both sweeps."""
    rows, eng = _sweep("em_quad12_T20", None, "cuda", 8192, 6, storage_dtype=torch.float32)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad") and eng.post.dtype == torch.float32
    b_med, b_p99, b_cost = MIXED_BOUNDS["em_quad12_T20"]
    for r in rows:
        assert r["mean_median"] <= b_med and r["mean_p99"] <= b_p99 and r["cost_median"] <= b_cost, r
        print("quad12 quad, fp32 storage, it %d: mean deviation median %.1e p99 %.1e max %.1e | cost median %.1e" %
              (r["it"], r["mean_median"], r["mean_p99"], r["mean"], r["cost_median"]))


@pytest.mark.gpu
def test_mixed_precision_wave_kernels_gpu():
    rows = _check_mixed("em_quad12_T20", None, "cuda", 1024, 6)
    for r in rows:
        print("quad12 wave, fp32 storage, it %d: mean deviation median %.1e p99 %.1e max %.1e | cost median %.1e" %
              (r["it"], r["mean_median"], r["mean_p99"], r["mean"], r["cost_median"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["em_pendulum_T200", "em_dcp_T300_run20"])
def test_fp32_arithmetic_first_iteration_bound_gpu(name):
    """What fp32 ARITHMETIC does at the full config shape: usable for exactly one EM iteration (bound asserted), then it
    leaves the fp64 run -- which is why the engine makes callers ask for it."""
    rows, _ = _sweep(name, None, "cuda", 4096, 4, dtype=torch.float32, allow_inexact=True)
    assert rows[0]["mean_median"] <= 5e-2, rows[0]
    assert max(r["mean_median"] for r in rows) > 1e-4  # documents that it is NOT parity-grade (if this ever fails: promote it)
    for r in rows:
        print(name, "fp32 arithmetic, it %d: mean deviation median %.1e max %.1e" % (r["it"], r["mean_median"], r["mean"]))
