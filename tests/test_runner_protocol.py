"""The reference runner's protocol (scripts/i2c_run.py:29-173) replayed on the DEVICE from committed fixtures.

tests/test_reference_runner.py executes the reference's own `run()` against this build, but only where the reference
checkout exists, hence only on the host simulation. Here the same sequence of calls is restated against the drop-in
package -- make_env / make_env_model, I2cGraph, the two policy classes, learn_msgs(), env.batch_eval (through i2c_rollout:
the simulator is attached to the graph), StochasticTrajectoryEvaluator, save_traj, the final noisy rollout -- and everything
`run()` leaves behind is compared with what the REFERENCE left behind for the same call (tests/golden/run_pendulum_seed0.npz,
run_pendulum_linearize_seed0.npz: captured by oracle/gen_golden.py case_i2c_run*). `-m gpu`: the HIP library on cuda:0."""
import os
import types

import numpy as np
import pytest

import hostsim
from golden_util import assert_close, load_case
from i2c.env import make_env
from i2c.exp_types import CubatureQuadrature, GaussianI2c, Linearize
from i2c.i2c import I2cGraph
from i2c.model import make_env_model
from i2c.policy.linear import ExpertTimeIndexedLinearGaussianPolicy, TimeIndexedLinearGaussianPolicy
from i2c.utils import StochasticTrajectoryEvaluator

N_EVAL = 10  # i2c_run.py:25


def _experiment(config):
    """scripts/experiments/pendulum_known_quad.py / pendulum_known.py as data (the modules cannot travel). The seed is set
    BEFORE the config draws its initial actions (i2c_run.py:215), so the NumPy stream is where the reference's was."""
    np.random.seed(0)
    T = 100
    if config == "pendulum_known_quad":
        inf = GaussianI2c(inference=CubatureQuadrature(1, 0, 0), Q=np.diag([1, 100.0, 1]), R=np.diag([2]), Qf=np.diag([1, 100.0, 1]),
                          alpha=100, alpha_update_tol=0.0, mu_u=1e-2 * np.random.randn(T, 1), sig_u=2.0 * np.eye(1),
                          mu_x_term=None, sig_x_term=None)
    else:
        inf = GaussianI2c(inference=Linearize(), Q=np.diag([1, 100.0, 1]), R=np.diag([1]), Qf=np.diag([1, 100.0, 1]),
                          alpha=100.0, alpha_update_tol=0.99, mu_u=np.zeros((T, 1)), sig_u=0.2 * np.eye(1),
                          mu_x_term=None, sig_x_term=None)
    return types.SimpleNamespace(ENVIRONMENT="PendulumKnown", MODEL=None, N_DURATION=T, N_INFERENCE=6,
                                 POLICY_COVAR=0.0 * np.eye(1), INFERENCE=inf)


def _replay(config, golden, lib, device, res_dir, tol):
    ref = load_case(golden)
    exp = _experiment(config)
    assert np.array_equal(exp.INFERENCE.mu_u, ref["mu_u"])  # the same initial actions as the reference's config import drew
    assert ref.meta["n_inference"] == exp.N_INFERENCE and ref.meta["T"] == exp.N_DURATION
    env = make_env(exp)
    model = make_env_model(exp.ENVIRONMENT, exp.MODEL)
    I = exp.INFERENCE
    i2c = I2cGraph(model, exp.N_DURATION, I.Q, I.R, I.Qf, I.alpha, I.alpha_update_tol, I.mu_u, I.sig_u, I.mu_x_term, I.sig_x_term,
                   I.inference, res_dir=res_dir, lib=lib, device=device)
    env.attach(i2c)  # batch_eval of policies written from this graph -> one i2c_rollout launch per call
    policy_linear = TimeIndexedLinearGaussianPolicy(exp.POLICY_COVAR, exp.N_DURATION, i2c.sys.dim_u, i2c.sys.dim_x)
    policy = ExpertTimeIndexedLinearGaussianPolicy(exp.POLICY_COVAR, exp.N_DURATION, i2c.sys.dim_u, i2c.sys.dim_x, soft=False)
    dim_terminal = i2c.Qf.shape[0]
    traj_eval = StochasticTrajectoryEvaluator(i2c.QR, i2c.Qf, i2c.z, i2c.z_term, dim_terminal)
    traj_eval_iter = StochasticTrajectoryEvaluator(i2c.QR, i2c.Qf, i2c.z, i2c.z_term, dim_terminal)
    traj_eval_safe = StochasticTrajectoryEvaluator(i2c.QR, i2c.Qf, i2c.z, i2c.z_term, dim_terminal)
    i2c.reset_metrics()
    assert env.simulated
    policy.zero()
    xs, ys, zs, z_term = env.batch_eval(policy, N_EVAL)  # a hand-made (zeroed) policy: host path, as in the reference
    traj_eval.eval(zs, z_term, zs[0], z_term[0])
    launches = []
    real_rollout = i2c.engine.rollout
    i2c.engine.rollout = lambda *a, **k: (launches.append(1), real_rollout(*a, **k))[1]
    for i in range(exp.N_INFERENCE):
        i2c.learn_msgs()
        policy_linear.write(*i2c.get_local_linear_policy())
        xs, ys, zs, zs_term = env.batch_eval(policy_linear, N_EVAL)
        assert len(xs) == N_EVAL and xs[0].shape == (exp.N_DURATION, 3) and np.all(np.isfinite(np.asarray(xs)))
        z_est, z_term_est = i2c.get_marginal_observed_trajectory()
        traj_eval_iter.eval(zs, zs_term, z_est, z_term_est)
        policy.write(*i2c.get_local_expert_linear_policy())
        xs, ys, zs, zs_term = env.batch_eval(policy, N_EVAL)
        traj_eval_safe.eval(zs, zs_term, z_est, z_term_est)
        if i == 0:
            xs, ys, zs, zs_term = env.batch_eval(policy, N_EVAL, deterministic=False)
    assert len(launches) == 2 * exp.N_INFERENCE + 1, "every evaluation of a graph-written policy ran as ONE device launch"
    assert len(traj_eval_iter.actual_cost_10) == exp.N_INFERENCE and np.all(np.isfinite(traj_eval_iter.actual_cost_90))
    policy_linear.write(*i2c.get_local_linear_policy())
    z_est, z_term_est = i2c.get_marginal_observed_trajectory()
    for _ in range(2):  # "evaluation stochastic" / "evaluation deterministic" (i2c_run.py:137-143)
        xs, ys, zs, zs_term = env.batch_eval(policy_linear, N_EVAL)
    traj_eval_iter.eval(zs, zs_term, z_est, z_term_est)
    traj_eval.eval(zs, zs_term, z_est, z_term_est)
    policy_linear.write(*i2c.get_local_linear_policy())
    x_final, y_final, _, _ = env.run(policy_linear)  # the noisy plant, NumPy's global stream (i2c_run.py:158)
    i2c.save_traj(res_dir)

    assert_close(np.array(i2c.costs_m), ref["costs_m"], tol, "costs_m")
    assert_close(np.array(i2c.alphas), ref["alphas"], tol, "alphas")
    assert_close(np.array(i2c.alphas_desired), ref["alphas_desired"], tol, "alphas_desired")
    K, k, sigK = i2c.get_local_linear_policy()
    assert_close(K, ref["K"], tol * 100, "K")
    assert_close(k, ref["k"], tol * 100, "k")
    assert_close(sigK, ref["sigK"], tol * 100, "sigK")
    scale = np.abs(ref["xu_plan"]).max()
    for name in ("xu_plan", "x_plan", "u_plan", "z_plan"):  # the files save_traj writes (i2c.py:1374-1382)
        mine = np.load(os.path.join(res_dir, name + ".npy"))
        assert mine.shape == ref[name].shape, (name, mine.shape, ref[name].shape)
        assert np.abs(mine - ref[name]).max() <= tol * 10 * max(scale, np.abs(ref[name]).max()), name
    nx = i2c.sys.dim_x
    real = {"xu_real": x_final, "dx_real": y_final, "x_real": x_final[:, :nx], "u_real": x_final[:, nx:]}  # i2c_run.py:176-184
    for name, mine in real.items():
        assert mine.shape == ref[name].shape, (name, mine.shape, ref[name].shape)
        assert_close(mine, ref[name], tol * 100, name)


CONFIGS = [("pendulum_known_quad", "run_pendulum_seed0"), ("pendulum_known", "run_pendulum_linearize_seed0")]


@pytest.mark.parametrize("config,golden", CONFIGS)
def test_runner_protocol_hostsim(tmp_path, config, golden):
    _replay(config, golden, hostsim.load(), "cpu", str(tmp_path), 1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("config,golden", CONFIGS)
def test_runner_protocol_gpu(tmp_path, config, golden):
    _replay(config, golden, None, "cuda", str(tmp_path), 1e-8)
