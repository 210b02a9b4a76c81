"""Row (f3): batched policy rollouts (`i2c_rollout`, replacing env.batch_eval's mp.Pool, i2c/env.py:93-103)
against an independent NumPy simulation with the same noise samples, and the cost evaluator."""
import numpy as np
import pytest
import torch

import hostsim
import parity
from golden_util import assert_close, load_case
from oracle.models_numpy import make_model


def _numpy_rollouts(om, post, x0, sig_x0, res, policy, nx, nu):
    """BaseSim.run (env.py:40-74) + linear.py policies, for every (r, b)."""
    K, k, sigK, mu, sig = post
    R_, B, T = res["xu"].shape[0], res["xu"].shape[1], res["xu"].shape[2]
    N = R_ * B
    ex0 = None if res["eps_x0"] is None else res["eps_x0"].numpy().reshape(nx, R_, B)
    ex = None if res["eps_x"] is None else res["eps_x"].numpy().reshape(T, nx, R_, B)
    eu = None if res["eps_u"] is None else res["eps_u"].numpy().reshape(T, nu, R_, B)
    Le = np.linalg.cholesky(om.sig_eta)
    xu_all = np.zeros((R_, B, T, nx + nu))
    z_all = np.zeros((R_, B, T, om.dim_z))
    xf_all = np.zeros((R_, B, nx))
    for r in range(R_):
        for b in range(B):
            x = x0[b].copy()
            if ex0 is not None:
                x = x + np.linalg.cholesky(sig_x0[b]) @ ex0[:, r, b]
            for t in range(T):
                if policy == "linear":
                    u = K[b, t] @ x + k[b, t]
                else:
                    d = x - mu[b, t, :nx]
                    e = 0.5 * d @ np.linalg.solve(sig[b, t, :nx, :nx], d)
                    w = np.exp(-e) if policy == "expert_soft" else float(abs(e) < 3.0)
                    u = mu[b, t, nx:] + w * (K[b, t] @ d)
                if eu is not None:
                    u = u + np.linalg.cholesky(sigK[b, t]) @ eu[t, :, r, b]
                xu = np.concatenate((x, u))
                xu_all[r, b, t] = xu
                z_all[r, b, t] = om.observe(xu[None])[0]
                x = om.dynamics(xu[None])[0]
                if ex is not None:
                    x = x + Le @ ex[t, :, r, b]
            xf_all[r, b] = x
    return xu_all, z_all, xf_all


def _check(lib, device, case, policy, B=3, R_=4, **noise):
    g = load_case(case)
    x0, mu_u = parity.batched_inputs(g, B)
    eng = parity.engine_from_case(g, lib, device, x0=x0, mu_u=mu_u)
    for _ in range(3):
        eng.learn_msgs()
    gen = torch.Generator(device=eng.device).manual_seed(7)
    res = eng.rollout(R_, policy, generator=gen, **noise)
    res = {k: (None if v is None else v.cpu().double()) for k, v in res.items()}
    K, k, sigK = (parity.np_(t) for t in eng.local_linear_policy())
    mu, sig = (parity.np_(t) for t in eng.marginal_state_action())
    om = make_model(g.meta["model"])
    sig_x0 = np.broadcast_to(om.sig_x0, (B,) + om.sig_x0.shape)
    xu, z, xf = _numpy_rollouts(om, (K, k, sigK, mu, sig), x0, sig_x0, res, policy, eng.nx, eng.nu)
    assert_close(res["xu"].numpy(), xu, 1e-9, f"{case} {policy} xu")
    assert_close(res["z"].numpy(), z, 1e-9, f"{case} {policy} z")
    assert_close(res["x_final"].numpy(), xf, 1e-9, f"{case} {policy} x_final")
    if res["z_term"] is not None:
        assert_close(res["z_term"].numpy(), om.observe_terminal(xf), 1e-9, f"{case} {policy} z_term")
    # different rollouts of one trajectory see different noise; with no noise at all they coincide
    if noise.get("process_noise", True):
        assert np.abs(xu[0] - xu[1]).max() > 0


CASES = [
    ("em_pendulum_T40_quad_general", "linear", dict(process_noise=True)),
    ("em_pendulum_T40_quad_general", "expert_soft", dict(process_noise=True, action_noise=True)),
    ("em_pendulum_T40_quad_general", "expert_hard", dict(process_noise=False)),
    ("em_dcp_T60", "linear", dict(process_noise=True, sample_x0=True)),
    ("em_quadrotor_T20", "expert_soft", dict(process_noise=True, action_noise=True)),
]


@pytest.mark.parametrize("case,policy,noise", CASES)
def test_rollout_matches_numpy_simulation_cpu(case, policy, noise):
    _check(hostsim.load(), "cpu", case, policy, **noise)


@pytest.mark.gpu
@pytest.mark.parametrize("case,policy,noise", CASES)
def test_rollout_matches_numpy_simulation_gpu(case, policy, noise):
    _check(None, "cuda", case, policy, **noise)


def test_env_batch_eval_uses_the_device_and_matches_host_protocol():
    """i2c_run.py's evaluation calls (i2c_run.py:98-106) against the simulator mirror."""
    from i2c.env import make_env
    from i2c.exp_types import CubatureQuadrature
    from i2c.i2c import I2cGraph
    from i2c.model import make_env_model
    from i2c.policy.linear import ExpertTimeIndexedLinearGaussianPolicy, TimeIndexedLinearGaussianPolicy
    from i2c.utils import StochasticTrajectoryEvaluator

    class Exp:
        ENVIRONMENT, N_DURATION = "PendulumKnown", 30

    g = load_case("em_pendulum_T40_quad_general")
    env = make_env(Exp)
    model = make_env_model(Exp.ENVIRONMENT, None)
    i2c = I2cGraph(model, 30, g["Q"], g["R"], g["Qf"], 100.0, 0.0, g["mu_u"][:30], g["sig_u"], None, None,
                   CubatureQuadrature(1, 0, 0), lib=hostsim.load(), device="cpu")
    for _ in range(4):
        i2c.learn_msgs()
    env.attach(i2c)
    lin = TimeIndexedLinearGaussianPolicy(np.zeros((1, 1)), 30, 1, 2)
    lin.write(*i2c.get_local_linear_policy())
    xs, ys, zs, zts = env.batch_eval(lin, 10)
    assert len(xs) == 10 and xs[0].shape == (30, 3) and ys[0].shape == (30, 2) and zs[0].shape == (30, 4)
    assert zts[0].shape == (1, 3)
    assert_close(ys[0][:-1], xs[0][1:, :2] - xs[0][:-1, :2], 1e-12, "dx = x_{t+1} - x_t")
    # the same policy stepped on the host (reference protocol) in a noise-free environment equals a
    # noise-free device rollout
    env.deterministic = True
    x_host, _, z_host, zt_host = env.run(lin)
    xs_d, _, zs_d, zts_d = env.batch_eval(lin, 2)
    assert_close(xs_d[0], x_host, 1e-10, "device vs host rollout")
    assert_close(zs_d[1], z_host, 1e-10)
    assert_close(zts_d[0], zt_host, 1e-10)
    exp_pol = ExpertTimeIndexedLinearGaussianPolicy(np.zeros((1, 1)), 30, 1, 2, soft=False)
    exp_pol.write(*i2c.get_local_expert_linear_policy())
    xs_e, _, _, _ = env.batch_eval(exp_pol, 2)
    x_host_e, _, _, _ = env.run(exp_pol)
    assert_close(xs_e[0], x_host_e, 1e-10, "expert policy: device vs host")
    # a policy that no longer matches the graph falls back to the host protocol
    lin.k = lin.k + 1.0
    xs2, _, _, _ = env.batch_eval(lin, 1)
    assert np.abs(xs2[0] - x_host).max() > 1e-3
    ev = StochasticTrajectoryEvaluator(i2c.QR, i2c.Qf, i2c.z, i2c.z_term, i2c.Qf.shape[0])
    z_est, z_term_est = i2c.get_marginal_observed_trajectory()
    ev.eval(zs, zts, z_est, z_term_est)
    assert ev.actual_cost_10[0] <= ev.mu_actual_cost[0] <= ev.actual_cost_90[0] or len(set(np.round(ev.mu_actual_cost, 6))) == 1


# ---- pinned to the REFERENCE (tests/golden/rollouts_pendulum_T60.npz, oracle/gen_golden.py::case_rollouts): its BaseSim.run on
# its own PendulumKnown simulator with its own policy classes filled from its I2cGraph, and its StochasticTrajectoryEvaluator --
def _reference_rollout_case(lib, device):
    g = load_case("rollouts_pendulum_T60")
    eng = parity.engine_from_case(g, lib, device)
    for _ in range(g.meta["n_em"]):
        eng.learn_msgs()
    K, k, sigK = (parity.np_(t)[0] for t in eng.local_linear_policy())
    assert_close(K, g["K"], 1e-7, "controller the rollouts use")
    return g, eng


def _cmp(res, g, tag, r=0, tol=1e-7):
    xu = res["xu"][r, 0].cpu().double().numpy()
    xf = res["x_final"][r, 0].cpu().double().numpy()
    assert_close(xu, g[tag + "/xu"], tol, tag + " xu")
    assert_close(res["z"][r, 0].cpu().double().numpy(), g[tag + "/z"], tol, tag + " z")
    assert_close(res["z_term"][r, 0].cpu().double().numpy(), g[tag + "/z_term"], tol, tag + " z_term")
    x_all = np.concatenate((xu[:, :2], xf[None]), axis=0)
    assert_close(x_all[1:] - x_all[:-1], g[tag + "/dx"], tol * 10, tag + " dx")


def _rollouts_vs_reference(lib, device):
    g, eng = _reference_rollout_case(lib, device)
    for name in ("linear", "expert_soft", "expert_hard"):  # deterministic plant, deterministic policy: RNG-free
        _cmp(eng.rollout(1, name, process_noise=False, action_noise=False), g, "det/" + name)
    for name in ("linear", "expert_soft"):  # the reference's own noise draws, standardised, replayed as the kernel's eps
        ex, eu = g[f"sto/{name}/eps_x"], g[f"sto/{name}/eps_u"]  # (3, T, nx), (3, T, nu)
        res = eng.rollout(3, name, eps_x=np.ascontiguousarray(np.transpose(ex, (1, 2, 0))),
                          eps_u=np.ascontiguousarray(np.transpose(eu, (1, 2, 0))))
        for r in range(3):
            _cmp(res, g, f"sto/{name}/{r}", r=r, tol=1e-6)


def test_rollout_kernel_vs_reference_simulator_cpu():
    _rollouts_vs_reference(hostsim.load(), "cpu")


@pytest.mark.gpu
def test_rollout_kernel_vs_reference_simulator_gpu():
    _rollouts_vs_reference(None, "cuda")


def test_host_protocol_and_evaluator_vs_reference():
    """The mirrors a runner script touches -- env.run, the policy classes, StochasticTrajectoryEvaluator -- against the
    reference's captured outputs (deterministic runs exactly; the evaluator on the reference's own stochastic rollouts)."""
    from i2c.env import make_env
    from i2c.policy.linear import ExpertTimeIndexedLinearGaussianPolicy, TimeIndexedLinearGaussianPolicy
    from i2c.utils import StochasticTrajectoryEvaluator

    g = load_case("rollouts_pendulum_T60")
    T = g.meta["T"]

    class Exp:
        ENVIRONMENT, N_DURATION = "PendulumKnown", T

    env = make_env(Exp)
    env.deterministic = True
    lin = TimeIndexedLinearGaussianPolicy(g["sig_u"], T, 1, 2)
    lin.write(g["K"], g["k"], g["sigK"])
    pols = {"linear": lin}
    for soft in (True, False):
        pe = ExpertTimeIndexedLinearGaussianPolicy(g["sig_u"], T, 1, 2, soft=soft)
        pe.write(g["expert/K"], g["expert/k"], g["expert/sigK"], g["expert/mu"], g["expert/lam"])
        pols["expert_soft" if soft else "expert_hard"] = pe
    for name, pol in pols.items():
        xt, yt, zt, z_term = env.run(pol, deterministic=True)
        assert_close(xt, g[f"det/{name}/xu"], 1e-12, name + " xu")
        assert_close(yt, g[f"det/{name}/dx"], 1e-10, name + " dx")
        assert_close(zt, g[f"det/{name}/z"], 1e-12, name + " z")
        assert_close(np.reshape(z_term, -1), g[f"det/{name}/z_term"], 1e-12, name + " z_term")
    Q, R = g["Q"], g["R"]
    QR = np.zeros((4, 4))
    QR[:3, :3], QR[3:, 3:] = Q, R
    zg = np.array([0.0, 1.0, 0.0, 0.0])  # PendulumKnown.zg (env_def.py:262-266): sin, cos, thd, u of the upright pendulum
    ev = StochasticTrajectoryEvaluator(QR, g["Qf"], zg, zg[:3], 3)
    zs = [g[f"sto/linear/{r}/z"] for r in range(3)]
    zts = [g[f"sto/linear/{r}/z_term"].reshape(1, -1) for r in range(3)]
    ev.eval(zs, zts, g["eval/z_est"], g["eval/z_term_est"])
    ev.eval(zs[:2], zts[:2], g["eval/z_est"], g["eval/z_term_est"])
    for key in ("mu_actual_cost", "min_actual_cost", "max_actual_cost", "actual_cost_10", "actual_cost_90", "planned_cost"):
        assert_close(np.asarray(getattr(ev, key), float).reshape(-1), g["eval/" + key], 1e-12, "evaluator " + key)
