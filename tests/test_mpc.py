"""Row (f2): the MPC loop either side of the solver -- batched cubature Kalman filter
(i2c_ckf_filter), warm-started sweeps, receding-horizon shift -- replayed against what the
REFERENCE's PartiallyObservedMpcPolicy (i2c/policy/mpc.py:113-182) did on the same measurement
stream (tests/golden/mpc_*.npz, oracle/gen_golden.py). Protocol of mpc_quad.py:624-650:
calibrate_alpha, warm start, calibrate_alpha, closed loop with per-cell moving targets."""
import numpy as np
import pytest
import torch

import hostsim
import parity
from golden_util import assert_close, load_case
from i2c.exp_types import CubatureQuadrature
from i2c.i2c import I2cGraph
from i2c.policy.mpc import PartiallyObservedMpcPolicy


def _rule_of(meta):
    from i2c.exp_types import GaussHermiteQuadrature, Linearize

    kind = meta.get("inference", "cubature")
    return {"cubature": lambda: CubatureQuadrature(*meta["quad"]), "linearize": Linearize,
            "gauss_hermite": lambda: GaussHermiteQuadrature(meta.get("gh_degree", 3))}[kind]()


def _policy(g, lib, device, batch=None, group_lanes=0, rule=None):
    meta = g.meta
    model = parity.product_model(g)
    model.sig_zeta = g["sig_zeta"]
    i2c = I2cGraph(model, meta["T"], g.get("Q"), g["R"], g.get("Qf"), meta["alpha"], meta["tol"], g["mu_u"], g["sig_u"],
                   None, None, _rule_of(meta) if rule is None else rule, lib=lib, device=device, batch=batch,
                   group_lanes=group_lanes)
    i2c._propagate = True
    pol = PartiallyObservedMpcPolicy(i2c, meta["n_iter"], g["sig_u"], np.copy(g["z_traj"]))
    pol.set_control(feedforward=meta["feedforward"])
    return model, i2c, pol


def _replay(name, lib, device, tol, group_lanes=0):
    g = load_case(name)
    meta = g.meta
    model, i2c, pol = _policy(g, lib, device, group_lanes=group_lanes)
    i2c.calibrate_alpha()
    assert_close(i2c.alpha, g["alpha_cal1"], tol, "first alpha calibration")
    pol.optimize(meta["warm"], model.x0, model.sig_x0)
    i2c.calibrate_alpha()
    assert_close(i2c.alpha, g["alpha_cal2"], tol, "second alpha calibration")
    lo, hi = model.xu_lim[0, model.dim_x:], model.xu_lim[1, model.dim_x:]
    for t in range(meta["steps"]):
        u = pol(t, g["y"][t].reshape(-1, 1), g["u_prev"][t].reshape(-1, 1))
        u = np.clip(u.T, lo, hi).T
        assert u.shape == (model.dim_u, 1) and pol.mu.shape == (model.dim_x, 1)
        assert_close(pol.mu[:, 0], g["mu"][t], tol, f"{name} belief mean, step {t}")
        assert_close(pol.covar, g["covar"][t], tol * 10, f"{name} belief covariance, step {t}")
        assert_close(u[:, 0], g["ctrl"][t], tol * 10, f"{name} control, step {t}")
        assert_close(i2c.alpha, g["alpha_steps"][t], tol, f"{name} alpha, step {t}")
    assert_close(pol.xu_history[-1][:, :, 0], g["xu_plan_last"], tol * 10, f"{name} last plan")
    # the histories of the single closed loop (round 6: device-side snapshots, host arrays when read): one entry per step, the
    # filtered beliefs the steps reported, every plan with the reference's shape
    assert len(pol.mus) == len(pol.covars) == len(pol.xu_history) == len(pol.z_history) == meta["steps"]
    assert_close(np.stack([m[:, 0] for m in pol.mus]), g["mu"], tol, f"{name} mus history")
    assert_close(np.stack(list(pol.covars)), g["covar"], tol * 10, f"{name} covars history")
    assert all(h.shape == (meta["T"], model.dim_xu, 1) for h in pol.xu_history[:3]) and pol.z_history[0].shape == (meta["T"], model.dim_z)
    # the horizon shift: after the loop the last cell is a fresh feed-forward cell
    assert i2c.cells[-1].state_action_independence
    return pol


@pytest.mark.parametrize("name", ["mpc_pendulum_ff", "mpc_pendulum_fb", "mpc_pendulum_fb_general", "mpc_pendulum_fb_lin", "mpc_pendulum_ff_lin", "mpc_pendulum_fb_gh3",
                                  "mpc_pendulum_fb_short_targets", "mpc_pendulum_random0", "mpc_pendulum_random1", "mpc_pendulum_random2",
                                  "mpc_pendulum_random3", "mpc_pendulum_random4", "mpc_pendulum_random5",
                                  "mpc_quadrotor_fb", "mpc_quad12_fb", "mpc_quad12_fb_lin"])
def test_mpc_replay_cpu(name):
    _replay(name, hostsim.load(), "cpu", 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mpc_pendulum_ff", "mpc_pendulum_fb", "mpc_pendulum_fb_general", "mpc_pendulum_fb_lin", "mpc_pendulum_ff_lin", "mpc_pendulum_fb_gh3",
                                  "mpc_pendulum_fb_short_targets", "mpc_pendulum_random0", "mpc_pendulum_random1", "mpc_pendulum_random2",
                                  "mpc_pendulum_random3", "mpc_pendulum_random4", "mpc_pendulum_random5",
                                  "mpc_quadrotor_fb", "mpc_quad12_fb", "mpc_quad12_fb_lin"])
def test_mpc_replay_gpu(name):
    _replay(name, None, "cuda", 1e-6)


# the reference's MPC loops with the forward sweeps on the quad kernel (csrc/i2c_quad.hpp: per-cell targets and temperatures,
# the ring of per-cell buffers, a terminal cell that sits mid-horizon after the first shift)
@pytest.mark.parametrize("name", ["mpc_pendulum_ff", "mpc_pendulum_fb", "mpc_pendulum_fb_short_targets", "mpc_pendulum_random0",
                                  "mpc_pendulum_random5", "mpc_quadrotor_fb"])
def test_mpc_replay_quad_forward_cpu(name):
    pol = _replay(name, hostsim.load(), "cpu", 1e-7, group_lanes=64)
    assert pol.engine.forward_family == "quad"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mpc_pendulum_ff", "mpc_pendulum_fb", "mpc_pendulum_fb_short_targets", "mpc_pendulum_random0",
                                  "mpc_pendulum_random5", "mpc_quadrotor_fb"])
def test_mpc_replay_quad_forward_gpu(name):
    pol = _replay(name, None, "cuda", 1e-6, group_lanes=64)
    assert pol.engine.forward_family == "quad"


# BASELINE config 4 at its FULL horizon (H = 50, two EM iterations per control step, mpc_quad.py:559), pinned to the reference's own
# PartiallyObservedMpcPolicy (oracle/gen_golden.py: mpc_quad12_H50, mpc_quad_H50): the single loop, and a batch of closed loops
# whose lane 0 carries the reference's problem (the batch runs the one-call control step i2c_mpc_step; at B >= 2048 the 12-state
# model's forward sweeps are the quad kernel's)
@pytest.mark.parametrize("name", ["mpc_quad12_fb_H50", "mpc_quadrotor_fb_H50"])
def test_mpc_replay_full_horizon_cpu(name):
    _replay(name, hostsim.load(), "cpu", 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name,lanes", [("mpc_quad12_fb_H50", 0), ("mpc_quad12_fb_H50", 16), ("mpc_quadrotor_fb_H50", 0), ("mpc_quadrotor_fb_H50", -1)])
def test_mpc_replay_full_horizon_gpu(name, lanes):
    _replay(name, None, "cuda", 1e-6, group_lanes=lanes)


def _replay_batched(name, lib, device, tol, B, group_lanes=0):
    """B closed loops at once, lane 0 = the reference's problem (its measurement stream from the fixture), the other lanes
    perturbed copies driven by their own simulated plants: lane 0's belief / control / temperature against the fixture."""
    g = load_case(name)
    meta = g.meta
    model = parity.product_model(g)
    model.sig_zeta = g["sig_zeta"]
    rng = np.random.default_rng(3)
    nx, nu = model.dim_x, model.dim_u
    x0 = np.tile(np.asarray(model.x0, float).reshape(1, -1), (B, 1)) + 1e-3 * rng.normal(size=(B, nx))
    x0[0] = np.asarray(model.x0, float).reshape(-1)
    i2c = I2cGraph(model, meta["T"], g.get("Q"), g["R"], g.get("Qf"), meta["alpha"], meta["tol"], g["mu_u"], g["sig_u"], None, None,
                   _rule_of(meta), lib=lib, device=device, batch=B, x0=x0, group_lanes=group_lanes)
    i2c._propagate = True
    pol = PartiallyObservedMpcPolicy(i2c, meta["n_iter"], g["sig_u"], np.copy(g["z_traj"]))
    pol.set_control(feedforward=meta["feedforward"])
    e = i2c.engine
    i2c.calibrate_alpha()
    assert_close(parity.np_(e.alpha)[0], g["alpha_cal1"], tol, "first alpha calibration, lane 0")
    pol.optimize(meta["warm"])
    i2c.calibrate_alpha()
    assert_close(parity.np_(e.alpha)[0], g["alpha_cal2"], tol, "second alpha calibration, lane 0")
    lo, hi = model.xu_lim[0, nx:], model.xu_lim[1, nx:]
    x = x0.copy()
    y = model.measure(x)
    u = np.zeros((B, nu))
    for t in range(meta["steps"]):
        y[0], u[0] = g["y"][t], g["u_prev"][t]
        ub = np.clip(pol(t, y, u), lo, hi)
        assert ub.shape == (B, nu) and np.all(np.isfinite(ub))
        assert_close(pol.mu[0], g["mu"][t], tol, f"{name} B={B} belief mean of lane 0, step {t}")
        assert_close(pol.covar[0], g["covar"][t], tol * 10, f"{name} B={B} belief covariance of lane 0, step {t}")
        assert_close(ub[0], g["ctrl"][t], tol * 10, f"{name} B={B} control of lane 0, step {t}")
        assert_close(parity.np_(e.alpha)[0], g["alpha_steps"][t], tol, f"{name} B={B} alpha of lane 0, step {t}")
        u = ub
        x = model.dynamics(np.hstack((x, u)))  # the other lanes' plants (noise-free)
        y = model.measure(x)
    assert e.failures() == [] and e.t0 == meta["steps"] % meta["T"]
    return pol


def test_mpc_replay_full_horizon_batched_cpu():
    pol = _replay_batched("mpc_quadrotor_fb_H50", hostsim.load(), "cpu", 1e-7, 5)
    assert pol.engine.forward_family == "quad"


def test_mpc_replay_full_horizon_quad12_quad_sweeps_cpu():
    """Config 4's closed loop with BOTH sweeps on the quad kernels (the default from 4096 trajectories up; asked for here): ring
    of per-cell buffers, per-cell targets and temperatures, one-call control step, a ragged batch of three loops."""
    pol = _replay_batched("mpc_quad12_fb_H50", hostsim.load(), "cpu", 1e-7, 3, group_lanes=parity.pkg._native.LANES_QUAD)
    assert (pol.engine.forward_family, pol.engine.backward_family) == ("quad", "quad")


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,fam", [("mpc_quad12_fb_H50", 8192, "quad"), ("mpc_quad12_fb_H50", 1024, "wave"), ("mpc_quadrotor_fb_H50", 8192, "quad"),
                                        ("mpc_quadrotor_fb_H50", 1024, "quad")])
def test_mpc_replay_full_horizon_batched_gpu(name, B, fam):
    """BASELINE config 4's shapes (B = 8192: the whole config on one GPU; 1024: one GPU's share of it), against the reference."""
    pol = _replay_batched(name, None, "cuda", 1e-6, B)
    assert pol.engine.forward_family == fam
    if name == "mpc_quad12_fb_H50":  # (the 12-state model's backward sweep: quad above 2048 trajectories, wave up to there)
        assert pol.engine.backward_family == ("quad" if B > 2048 else "wave")


# mpc_quad12_fb above runs the 12-state quadrotor's default, the wave kernels; the same replay on its group kernels
def test_mpc_replay_quad12_group_kernels_cpu():
    pol = _replay("mpc_quad12_fb", hostsim.load(), "cpu", 1e-7, group_lanes=16)
    assert pol.engine.forward_family == "group"


@pytest.mark.gpu
def test_mpc_replay_quad12_group_kernels_gpu():
    pol = _replay("mpc_quad12_fb", None, "cuda", 1e-6, group_lanes=16)
    assert pol.engine.forward_family == "group"


def _ckf_oracle(model, rule_w, mu, cov, u, y, sig_zeta):
    """PartiallyObservedMpcPolicy.filter (mpc.py:125-145), batched, with the oracle's transform."""
    from oracle.i2c_numpy import CubatureRule, SigmaPointTransform

    tf = SigmaPointTransform(CubatureRule(*rule_w), model.dim_x)
    X = tf.points(mu, cov)
    XU = np.concatenate((X, np.broadcast_to(u[:, None, :], X.shape[:2] + (u.shape[-1],))), axis=-1)
    Xf = model.dynamics(XU)
    w = tf.w
    mu_f = np.einsum("p,bpi->bi", w, Xf)
    sig_f = np.einsum("p,bpi,bpj->bij", w, Xf, Xf) - mu_f[:, :, None] * mu_f[:, None, :] + w.sum() * model.sig_eta
    mu_y, sig_y, sig_xy, _, _ = tf.forward(model.measure, mu_f, sig_f)
    sig_y = sig_y + sig_zeta
    K = np.swapaxes(np.linalg.solve(np.swapaxes(sig_y, -1, -2), np.swapaxes(sig_xy, -1, -2)), -1, -2)
    return mu_f + np.einsum("bij,bj->bi", K, y - mu_y), sig_f - K @ sig_y @ np.swapaxes(K, -1, -2)


def _ckf_batch(lib, device, model_name, B=37, rule=None, group_lanes=0, family=None):
    """The filter kernel on a ragged batch of random beliefs against the oracle restatement. `rule`: the graph's inference --
    the filter's own rule is CubatureQuadrature(1, 0, 0) whatever the graph infers with (mpc.py:121-123)."""
    from oracle.models_numpy import make_model

    g = load_case({"PlanarQuadrotor": "mpc_quadrotor_fb", "Quadrotor12": "mpc_quad12_fb"}.get(model_name, "mpc_pendulum_ff"))
    rng = np.random.default_rng(3)
    om = make_model(model_name)
    nx, nu = om.dim_x, om.dim_u
    ny = om.measure(np.zeros((1, nx))).shape[-1]
    mu = np.asarray(om.x0) + 0.1 * rng.normal(size=(B, nx))
    A = rng.normal(size=(B, nx, nx))
    cov = 1e-3 * (A @ np.swapaxes(A, -1, -2)) + 1e-5 * np.eye(nx)
    u = rng.normal(size=(B, nu)) + (om.u_max / 4 if hasattr(om, "u_max") else 0.0)
    sig_zeta = np.diag(10.0 ** rng.uniform(-5, -2, size=ny))
    y = om.measure(mu) + 0.01 * rng.normal(size=(B, ny))
    _, i2c, pol = _policy(g, lib, device, batch=B, rule=rule, group_lanes=group_lanes)
    if family is not None:
        assert i2c.engine.kernel_family("filter") == family
    i2c.sys.sig_zeta = sig_zeta
    i2c.engine.set_initial_state(mu, cov)
    pol.filter(y, u)
    mu_o, cov_o = _ckf_oracle(om, (1, 0, 0), mu, cov, u, y, sig_zeta)
    assert_close(pol.mu, mu_o, 1e-9, "filtered mean")
    assert_close(pol.covar, cov_o, 1e-8, "filtered covariance")
    assert i2c.engine.failures() == []


@pytest.mark.parametrize("model_name", ["PendulumKnown", "PlanarQuadrotor", "Quadrotor12"])
def test_ckf_batch_cpu(model_name):
    _ckf_batch(hostsim.load(), "cpu", model_name, family="quad" if model_name == "Quadrotor12" else "lane")  # (round 5: the quad filter step for d = 16)


def test_ckf_batch_quad12_group_kernels_cpu():
    """The filter of the 12-state quadrotor on its group kernels (the fallback, asked for): the same oracle."""
    _ckf_batch(hostsim.load(), "cpu", "Quadrotor12", B=6, group_lanes=16, family="group")


@pytest.mark.gpu
def test_ckf_batch_quad12_group_kernels_gpu():
    _ckf_batch(None, "cuda", "Quadrotor12", group_lanes=16, family="group")


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["PendulumKnown", "PlanarQuadrotor", "Quadrotor12"])
def test_ckf_batch_gpu(model_name):
    _ckf_batch(None, "cuda", model_name, family="quad" if model_name == "Quadrotor12" else "lane")


def _graph_rules():
    from i2c.exp_types import GaussHermiteQuadrature, Linearize

    return {"general_weights": CubatureQuadrature(1.2, 0.44, 0.5), "linearize": Linearize(), "gauss_hermite": GaussHermiteQuadrature(3)}


CKF_RULES = [("PendulumKnown", "general_weights"), ("PendulumKnown", "linearize"), ("PendulumKnown", "gauss_hermite"),
             ("Quadrotor12", "general_weights"), ("Quadrotor12", "linearize")]


@pytest.mark.parametrize("model_name,rule", CKF_RULES)
def test_ckf_rule_is_fixed_cpu(model_name, rule):
    """The estimator under a graph that infers with another rule (group kernels for the 12-state quadrotor)."""
    _ckf_batch(hostsim.load(), "cpu", model_name, B=5, rule=_graph_rules()[rule])


@pytest.mark.gpu
@pytest.mark.parametrize("model_name,rule", CKF_RULES)
def test_ckf_rule_is_fixed_gpu(model_name, rule):
    _ckf_batch(None, "cuda", model_name, B=5, rule=_graph_rules()[rule])


def _batched_loop_matches_single(lib, device):
    """B closed loops driven at once: every lane reproduces the B = 1 loop on the same stream."""
    g = load_case("mpc_pendulum_fb")
    meta = g.meta
    B = 5
    _, i1, p1 = _policy(g, lib, device)
    _, iB, pB = _policy(g, lib, device, batch=B)
    for pol, i2c in ((p1, i1), (pB, iB)):
        i2c.calibrate_alpha()
        pol.optimize(meta["warm"], g["x0"], g["sig_x0"])
        i2c.calibrate_alpha()
    for t in range(6):
        y, u = g["y"][t].reshape(-1, 1), g["u_prev"][t].reshape(-1, 1)
        u1 = p1(t, y, u)
        uB = pB(t, np.tile(y.T, (B, 1)), np.tile(u.T, (B, 1)))
        assert uB.shape == (B, 1)
        assert_close(uB, np.tile(u1.T, (B, 1)), 1e-12, f"batched control, step {t}")
        assert_close(pB.mu, np.tile(p1.mu.T, (B, 1)), 1e-12, f"batched belief, step {t}")


def test_batched_mpc_loop_cpu():
    _batched_loop_matches_single(hostsim.load(), "cpu")


@pytest.mark.gpu
def test_batched_mpc_loop_gpu():
    _batched_loop_matches_single(None, "cuda")


def test_quadrature_inference_mirror_matches_reference_vectors():
    """The host-side QuadratureInference class that callers use directly (mpc_quad.py:129-152)."""
    from i2c.inference.quadrature import QuadratureInference
    from i2c.model import make_env_model

    q = load_case("quadrature_vectors")
    model = make_env_model("PendulumKnown")
    for n in range(int(q["n"])):
        pre = f"q{n}/"
        qi = QuadratureInference(CubatureQuadrature(*q[pre + "quad"]), 3)
        m, S = q[pre + "m"].reshape(-1, 1), q[pre + "S"]
        m_z, S_z = qi.forward(model.observe, m, S)
        assert m_z.shape == (4, 1)
        assert_close(qi.x_pts, q[pre + "x_pts"], 1e-14)
        assert_close(m_z[:, 0], q[pre + "obs_m"], 1e-13)
        assert_close(S_z, q[pre + "obs_S"], 1e-12)
        assert_close(qi.sig_xy, q[pre + "obs_Sxy"], 1e-12)
        m_y, S_y, S_n = qi.forward_gaussian(model.forward, m, S)
        assert_close(m_y[:, 0], q[pre + "dyn_m"], 1e-13)
        assert_close(S_y, q[pre + "dyn_S"], 1e-11)
        assert_close(S_n, q[pre + "dyn_noise"], 1e-15)


def _native_step_equals_stepwise(name, lib, device):
    """i2c_mpc_step (one library call per control step) against the same step made of separate calls (filter, sweeps,
    _update_priors, read-out, shift_horizon), which the replays above pin to the reference: bit-identical buffers."""
    g = load_case(name)
    meta = g.meta
    B = 3
    engines = []
    for _ in range(2):
        _, i2c, pol = _policy(g, lib, device, batch=B)
        i2c.calibrate_alpha()
        pol.optimize(meta["warm"], g["x0"] if "x0" in g else i2c.sys.x0, g["sig_x0"] if "sig_x0" in g else i2c.sys.sig_x0)
        i2c.calibrate_alpha()
        engines.append((i2c.engine, pol))
    (ea, pa), (eb, pb) = engines
    nz = ea.nz
    for t in range(5):
        y = np.tile(g["y"][t].reshape(1, -1), (B, 1))
        u = np.tile(g["u_prev"][t].reshape(1, -1), (B, 1))
        yd, ud = pa._dev(y, ea.dims.ny), pa._dev(u, ea.nu)
        z_new = pa._next_target(t)
        # (a) separate calls
        if t > 0:
            ea.ckf_filter(yd, ud, pa.i2c.sys.sig_zeta)
        for _ in range(meta["n_iter"]):
            ea.forward_backward()
            ea.update_priors()
        mu_a = ea.cells(ea.post)[0, ea.nx: ea.d, :].T.clone()
        ea.shift_horizon(z_new)
        # (b) one call
        mu_b, _ = eb.mpc_step(meta["n_iter"], yd if t > 0 else None, ud if t > 0 else None, pb.i2c.sys.sig_zeta, z_new=z_new)
        assert torch.equal(mu_a, mu_b), f"{name} step {t}: first action"
        for key in ("post", "x0", "sig_x0", "feedforward", "alpha_cell", "z"):
            ta, tb = getattr(ea, key), getattr(eb, key)
            if ta is not None:
                assert torch.equal(ta, tb), f"{name} step {t}: {key}"
        assert ea.terminal_cell == eb.terminal_cell and nz == eb.nz and ea.t0 == eb.t0 == (t + 1) % ea.H


@pytest.mark.parametrize("name", ["mpc_pendulum_fb", "mpc_pendulum_fb_lin", "mpc_pendulum_fb_gh3", "mpc_quadrotor_fb", "mpc_quad12_fb"])
def test_native_mpc_step_cpu(name):
    _native_step_equals_stepwise(name, hostsim.load(), "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mpc_pendulum_fb", "mpc_pendulum_fb_lin", "mpc_pendulum_fb_gh3", "mpc_quadrotor_fb", "mpc_quad12_fb"])
def test_native_mpc_step_gpu(name):
    _native_step_equals_stepwise(name, None, "cuda")


def _ring_equals_unrolled(inference, lib, device, golden=None, family="lane"):
    """The receding-horizon ring (I2cProblem.t0) for EVERY inference rule: an engine whose horizon has been shifted k times
    must give, bit for bit, the sweeps of an engine holding the same cells in cell order with the ring at its origin.
    (The reference's MpcPolicy accepts any I2cGraph, mpc.py:16-33; round 2 refused t0 != 0 outside the cubature path.)"""
    from i2c.exp_types import GaussHermiteQuadrature, Linearize
    from i2c.model import make_env_model
    from i2c.policy.mpc import MpcPolicy

    rule = {"cubature": CubatureQuadrature(1, 0, 0), "gauss_hermite": GaussHermiteQuadrature(3), "linearize": Linearize()}[inference]
    T, B = 9, 3
    rng = np.random.default_rng(5)
    if golden is None:
        model = make_env_model("PendulumKnown")
        Q, R, Qf, alpha = np.diag([1.0, 100.0, 1.0]), np.diag([2.0]), np.diag([1.0, 100.0, 1.0]), 100.0
        x0 = np.array([np.pi, 0.0])
    else:  # the wider models (group / wave kernel families): cost weights and start state of a golden case
        from golden_util import load_case

        c = load_case(golden)
        model = make_env_model(c.meta["model"])
        Q, R, Qf, alpha, x0 = c["Q"], c["R"], c["Qf"], c.meta["alpha"], c["x0"]
    nx, nu, nz = model.dim_x, model.dim_u, model.dim_z
    mu_u = 1e-1 * rng.normal(size=(B, T, nu))
    x0 = x0 + 1e-2 * rng.normal(size=(B, nx))
    z_traj = np.tile(np.asarray(model.zg).reshape(1, -1), (T + 6, 1)) + 1e-2 * rng.normal(size=(T + 6, nz))
    sig_u = 0.5 * np.eye(nu)

    def graph():
        g = I2cGraph(model, T, Q, R, Qf, alpha, 0.5, mu_u, sig_u, None, None, rule, lib=lib, device=device, batch=B)
        g.engine.set_initial_state(x0, np.broadcast_to(1e-4 * np.eye(nx), (B, nx, nx)))
        return g

    ga, gb = graph(), graph()
    assert ga.engine.forward_family == family
    pa, pb = MpcPolicy(ga, 2, sig_u, z_traj), MpcPolicy(gb, 2, sig_u, z_traj)
    pa.set_control(feedforward=False)
    pb.set_control(feedforward=False)
    ea, eb = ga.engine, gb.engine
    ea.set_cell_expert(2, False)  # the per-cell expert flags ride on the ring too
    eb.set_cell_expert(2, False)
    for step in range(5):
        # b := a in cell order, ring at the origin
        for key in ("post", "z", "alpha_cell", "feedforward", "expert_cells"):
            getattr(eb, key).copy_(ea.cells(getattr(ea, key)))
        for key in ("x0", "sig_x0", "alpha", "temp"):
            getattr(eb, key).copy_(getattr(ea, key))
        eb.terminal_cell = ea.terminal_cell
        eb._problem.terminal_cell = int(ea.terminal_cell)
        assert eb.t0 == 0 and ea.t0 == step
        for e in (ea, eb):
            for _ in range(2):
                e.forward_backward()
                e.update_priors()
        assert ea.failures() == [] and eb.failures() == []
        assert torch.equal(ea.cells(ea.post), eb.post), f"{inference}: posterior after {step} shifts"
        assert torch.equal(ea.fwd, eb.fwd), f"{inference}: forward messages after {step} shifts"
        assert torch.equal(ea.term_stats, eb.term_stats)
        ea.shift_horizon(pa._next_target(step))
    # reset() returns to the snapshot, wherever the ring was when it was taken (a second policy on a moved ring)
    pc = MpcPolicy(ga, 2, sig_u)
    before = ea.cells(ea.post).clone()
    t0 = ea.t0
    assert t0 != 0
    ea.forward_backward(); ea.update_priors(); ea.shift_horizon()
    pc.reset()
    assert ea.t0 == t0 and torch.equal(ea.cells(ea.post), before)


@pytest.mark.parametrize("inference", ["cubature", "gauss_hermite", "linearize"])
def test_mpc_ring_every_inference_cpu(inference):
    _ring_equals_unrolled(inference, hostsim.load(), "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("inference", ["cubature", "gauss_hermite", "linearize"])
def test_mpc_ring_every_inference_gpu(inference):
    _ring_equals_unrolled(inference, None, "cuda")


RING_WIDE = [("cubature", "em_dcp_T60", "quad"), ("cubature", "em_quad12_T20", "wave"), ("linearize", "lin_quad12_T20", "wave"),
             ("linearize", "lin_dcp_T80", "lane")]


@pytest.mark.parametrize("inference,golden,family", RING_WIDE)
def test_mpc_ring_wide_models_cpu(inference, golden, family):
    """The same for the models served by the group and wave kernel families (and their Linearize variants)."""
    _ring_equals_unrolled(inference, hostsim.load(), "cpu", golden, family)


@pytest.mark.gpu
@pytest.mark.parametrize("inference,golden,family", RING_WIDE)
def test_mpc_ring_wide_models_gpu(inference, golden, family):
    _ring_equals_unrolled(inference, None, "cuda", golden, family)
