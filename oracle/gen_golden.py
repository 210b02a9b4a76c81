"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by importing the REAL reference
(/root/reference) in this container through oracle/ref_shim.py.  Run:

    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python oracle/gen_golden.py [case ...]

The .npz files hold only data (inputs + the reference's outputs); the reference itself never
travels. Every array is float64. Per-iteration detail is stored under keys "it{n}/<name>".
"""
import os
import sys
import json

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import numpy as np  # noqa: E402
from i2c.i2c import I2cGraph  # noqa: E402  (the reference)
from i2c.model import make_env_model  # noqa: E402
from i2c.exp_types import CubatureQuadrature, GaussHermiteQuadrature  # noqa: E402
from i2c.inference.quadrature import QuadratureInference  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

FWD_KEYS = ["mu_xu0_f", "sig_xu0_f", "mu_xu1_f", "sig_xu1_f", "mu_x3_f", "sig_x3_f", "J_dyn", "mu_z0_f", "sig_z0_f"]
BWD_KEYS = ["mu_xu0_m", "sig_xu0_m", "K", "k", "sigK", "mu_z0_m", "sig_z0_m", "mu_x3_m", "sig_x3_m"]
PF_KEYS = ["mu_xu0_pf", "sig_xu0_pf", "mu_x3_pf", "sig_x3_pf", "mu_z0_pf", "sig_z0_pf"]


def _stack(cells, key):
    arrs = [np.asarray(getattr(c, key), dtype=float) for c in cells]
    a = np.stack(arrs)
    if a.ndim == 3 and a.shape[-1] == 1 and key.startswith(("mu_", "k")):
        a = a[..., 0]  # (T, n, 1) column means -> (T, n)
    if key in ("mu_z0_f", "mu_z0_m", "mu_z0_pf") and a.ndim == 3:
        a = a.reshape(a.shape[0], -1)
    return a


def capture(g, keys):
    return {k: _stack(g.cells, k) for k in keys}


def run_em(g, n_detail, n_total, out, pre_propagate=False):
    """Replicates I2cGraph.learn_msgs (i2c.py:1238-1245) with capture points in between."""
    if pre_propagate:
        g.propagate()
    for it in range(1, n_total + 1):
        g.em_iter += 1
        g._forward_msgs()
        if it <= n_detail:
            for k, v in capture(g, FWD_KEYS).items():
                out[f"it{it}/{k}"] = v
        g._backward_msgs()
        if it <= n_detail:
            for k, v in capture(g, BWD_KEYS).items():
                out[f"it{it}/{k}"] = v
            c = g.cells[-1]
            if c.mu_z3_m is not None:
                out[f"it{it}/mu_z3_m"] = np.asarray(c.mu_z3_m, dtype=float).reshape(-1)
                out[f"it{it}/sig_z3_m"] = np.asarray(c.sig_z3_m, dtype=float)
        if g._propagate:
            g.propagate()
            if it <= n_detail:
                for k, v in capture(g, PF_KEYS).items():
                    out[f"it{it}/{k}"] = v
        g._maximize()
    out["alphas"] = np.asarray(g.alphas, dtype=float)
    out["alphas_desired"] = np.asarray(g.alphas_desired, dtype=float)
    out["costs_m"] = np.asarray(g.costs_m, dtype=float)
    out["costs_m_var"] = np.asarray(g.costs_m_var, dtype=float)
    out["costs_pf"] = np.asarray(g.costs_pf, dtype=float)
    if g._propagate:
        out["alphas_pf"] = np.asarray(g.alphas_pf, dtype=float)
        out["costs_pf_var"] = np.asarray(g.costs_pf_var, dtype=float)
    if len(g.kl_terms):
        out["kl_terms"] = np.asarray(g.kl_terms, dtype=float)
    K, k, sigK = g.get_local_linear_policy()
    out["final/K"], out["final/k"], out["final/sigK"] = K, k, sigK
    mu, sig = g.get_marginal_state_action_distribution()
    out["final/mu_xu0_m"], out["final/sig_xu0_m"] = mu, sig
    out["final/mu_z0_m"] = g.get_marginal_observed_trajectory()[0]


def problem_inputs(model_name, model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, mu_x_term, sig_x_term, quad, **extra):
    meta = dict(model=model_name, T=int(T), alpha=float(alpha), tol=float(tol), quad=list(map(float, quad)))
    meta.update(extra)
    d = {
        "meta": np.array(json.dumps(meta)),
        "R": np.asarray(R, float),
        "mu_u": np.asarray(mu_u, float),
        "sig_u": np.asarray(sig_u, float),
        "x0": np.asarray(model.x0, float).reshape(-1),
        "sig_x0": np.asarray(model.sig_x0, float),
        "sig_eta": np.asarray(model.sig_eta, float),
    }
    if Q is not None:
        d["Q"] = np.asarray(Q, float)
    if Qf is not None:
        d["Qf"] = np.asarray(Qf, float)
    if mu_x_term is not None:
        d["mu_x_term"] = np.asarray(mu_x_term, float)
        d["sig_x_term"] = np.asarray(sig_x_term, float)
    return d


def save(name, out):
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")


# --------------------------------------------------------------------------------------
def case_pendulum(T=200, n_detail=3, n_total=30, seed=0, quad=(1, 0, 0), name="em_pendulum_T200"):
    """scripts/experiments/pendulum_known_quad.py:22-33 hyper-parameters at horizon T."""
    np.random.seed(seed)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, CubatureQuadrature(*quad))
    out = problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, quad, seed=seed)
    run_em(g, n_detail, n_total, out)
    save(name, out)


def case_pendulum_long():
    case_pendulum(T=200, n_detail=0, n_total=200, seed=0, name="em_pendulum_T200_run200")


def case_pendulum_seeds():
    """free-running EM for other seeds of the initial action sequence (summary only)."""
    for seed in (1, 2):
        case_pendulum(T=200, n_detail=0, n_total=60, seed=seed, name=f"em_pendulum_T200_seed{seed}_run60")


def case_pendulum_general_weights():
    case_pendulum(T=40, n_detail=2, n_total=6, seed=3, quad=(1.2, 0.44, 0.5), name="em_pendulum_T40_quad_general")


def case_double_cartpole(T=60, n_detail=2, n_total=10, name="em_dcp_T60"):
    """scripts/experiments/double_cartpole_known_cq.py:23-39 hyper-parameters."""
    np.random.seed(0)
    mu_u = 1e-2 * np.random.randn(T, 1)
    sf = 1e-3
    Q = sf * np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0])
    R = sf * np.diag([0.1])
    model = make_env_model("DoubleCartpoleKnown", None)
    g = I2cGraph(model, T, Q, R, Q, 0.05, 0.99, mu_u, np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("DoubleCartpoleKnown", model, T, Q, R, Q, 0.05, 0.99, mu_u, np.eye(1), None, None, (1, 0, 0), seed=0)
    run_em(g, n_detail, n_total, out)
    save(name, out)


def case_double_cartpole_T300():
    case_double_cartpole(T=300, n_detail=0, n_total=20, name="em_dcp_T300_run20")


def case_double_cartpole_T300_long():
    """50 free-running iterations of the double cartpole at the full horizon."""
    case_double_cartpole(T=300, n_detail=0, n_total=50, name="em_dcp_T300_run50")


def case_cartpole(T=100, n_detail=2, n_total=10):
    """scripts/experiments/cartpole_known_quad.py:24-35."""
    np.random.seed(0)
    mu_u = 1e-3 * np.random.randn(T, 1)
    Q = np.diag([1.0, 1.0, 100.0, 10.0, 1.0])
    R = np.diag([1.0])
    model = make_env_model("CartpoleKnown", None)
    g = I2cGraph(model, T, Q, R, Q, 80.0, 0.0, mu_u, np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("CartpoleKnown", model, T, Q, R, Q, 80.0, 0.0, mu_u, np.eye(1), None, None, (1, 0, 0), seed=0)
    run_em(g, n_detail, n_total, out)
    save("em_cartpole_T100", out)


def case_linear(T=60, n_detail=2, n_total=10, noise=1e-4):
    """scripts/experiments/linear_known_quad.py:21-33 with non-degenerate noise."""
    mu_u = np.zeros((T, 1))
    Q, R = np.diag([10.0, 10.0]), np.diag([1.0])
    model = make_env_model("LinearKnown", None)
    model.sig_x0 = noise * np.eye(2)
    model.sig_eta = noise * np.eye(2)
    g = I2cGraph(model, T, Q, R, Q, 800.0, 0.0, mu_u, np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("LinearKnown", model, T, Q, R, Q, 800.0, 0.0, mu_u, np.eye(1), None, None, (1, 0, 0), noise=noise)
    run_em(g, n_detail, n_total, out)
    save("em_linear_T60", out)


def case_covariance_control(T=100, n_detail=3, n_total=30):
    """scripts/nonlinear_covariance_control.py:81-124 + experiments/pendulum_known_act_reg_quad.py."""
    mu_u = np.zeros((T, 1))
    R = np.diag([1.0])
    mu_xt, sig_xt = np.array([0.0, 0.0]), np.diag([1e-3, 1e-3])
    model = make_env_model("PendulumKnownActReg", None)
    g = I2cGraph(model, T, None, R, None, 300.0, 1.0, mu_u, 0.5 * np.eye(1), mu_xt, sig_xt, CubatureQuadrature(1, 0, 0))
    for c in g.cells:
        c.use_expert_controller = False
    g._propagate = True
    out = problem_inputs(
        "PendulumKnownActReg", model, T, None, R, None, 300.0, 1.0, mu_u, 0.5 * np.eye(1), mu_xt, sig_xt, (1, 0, 0),
        propagate=True, use_expert_controller=False,
    )
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save("em_covctrl_T100", out)


def case_covariance_control_terminal_cost(T=40, n_detail=3, n_total=8):
    """Covariance control (tempered terminal prior, i2c.py:548-559) on a graph that ALSO has a terminal cost: the terminal
    observation statistics (:565-570) come from the pinned smoothed state; expert controller on; KL terms (:1012-1019)."""
    np.random.seed(3)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    mu_xt, sig_xt = np.array([0.5, 0.0]), np.array([[2e-3, 5e-4], [5e-4, 1e-2]])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), mu_xt, sig_xt, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), mu_xt, sig_xt, (1, 0, 0),
                         propagate=True, use_expert_controller=True, seed=3)
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save("em_covctrl_qf_T40", out)


def case_propagate_expert(T=50, n_detail=3, n_total=5):
    """Closed-loop propagation with the expert (pdf-ratio scaled) controller, i2c.py:160-165,
    plus calibrate_alpha (i2c.py:895-911) before the EM loop, as mpc_quad.py:624-630 does."""
    np.random.seed(1)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = problem_inputs(
        "PendulumKnown", model, T, Q, R, Qf, 100.0, 0.5, mu_u, 2.0 * np.eye(1), None, None, (1, 0, 0),
        propagate=True, use_expert_controller=True, calibrate_first=True, seed=1,
    )
    g.calibrate_alpha()
    out["alpha_calibrated"] = np.asarray(g.alpha, float)
    run_em(g, n_detail, n_total, out)
    save("em_pendulum_T50_propagate", out)


def case_quadrature():
    """QuadratureInference.forward / forward_gaussian unit vectors (quadrature.py:15-58)."""
    rng = np.random.default_rng(7)
    out = {}
    model = make_env_model("PendulumKnown", None)
    n = 0
    for quad in [(1, 0, 0), (0.7, 2.0, 0.5), (1.3, 0.0, 1.0)]:
        for _ in range(4):
            A = rng.normal(size=(3, 3))
            S = A @ A.T * 10 ** rng.uniform(-4, 0) + 1e-6 * np.eye(3)
            m = rng.normal(size=(3, 1)) * np.array([[3.0], [2.0], [1.0]])
            qi = QuadratureInference(CubatureQuadrature(*quad), 3)
            m_z, S_z = qi.forward(model.observe, m, S)
            pre = f"q{n}/"
            out[pre + "quad"] = np.array(quad, float)
            out[pre + "m"], out[pre + "S"] = m[:, 0], S
            out[pre + "x_pts"] = qi.x_pts
            out[pre + "obs_m"], out[pre + "obs_S"], out[pre + "obs_Sxy"] = m_z[:, 0], S_z, qi.sig_xy
            m_y, S_y, S_n = qi.forward_gaussian(model.forward, m, S)
            out[pre + "dyn_m"], out[pre + "dyn_S"], out[pre + "dyn_Sxy"], out[pre + "dyn_noise"] = (
                m_y[:, 0], S_y, qi.sig_xy, S_n,
            )
            n += 1
    out["n"] = np.array(n)
    # Gauss-Hermite rule parameters (exp_types.py:52-68)
    for deg in (2, 3, 4):
        gh = GaussHermiteQuadrature(deg)
        out[f"gh{deg}/pts"] = gh.pts(3)
        sf, w, _ = gh.weights(3)
        out[f"gh{deg}/sf"], out[f"gh{deg}/w"] = np.array(sf), w
    qi = QuadratureInference(GaussHermiteQuadrature(3), 3)
    m = np.array([[0.3], [-0.2], [0.1]])
    S = np.array([[0.2, 0.05, 0.0], [0.05, 0.1, 0.01], [0.0, 0.01, 0.3]])
    m_z, S_z = qi.forward(model.observe, m, S)
    out["gh3/m"], out["gh3/S"], out["gh3/obs_m"], out["gh3/obs_S"], out["gh3/obs_Sxy"] = m[:, 0], S, m_z[:, 0], S_z, qi.sig_xy
    save("quadrature_vectors", out)


def case_models():
    """Known-model plugins: dynamics / observe / observe_terminal on random inputs."""
    rng = np.random.default_rng(11)
    out = {}
    for name in ["PendulumKnown", "PendulumKnownActReg", "CartpoleKnown", "DoubleCartpoleKnown", "LinearKnown",
                 "LinearKnownMinimumEnergy"]:
        m = make_env_model(name, None)
        xu = rng.normal(size=(32, m.dim_xu)) * 3.0
        xu[:4, -1] *= 10  # exercise the action clip
        out[name + "/xu"] = xu
        dyn, noise = m.forward(xu)
        out[name + "/dyn"] = dyn
        out[name + "/noise0"] = noise[0]
        out[name + "/obs"] = m.observe(xu)
        zt = m.observe_terminal_x(xu[:, : m.dim_x])
        if zt is not None:
            out[name + "/obs_term"] = zt
        out[name + "/x0"] = np.asarray(m.x0, float).reshape(-1)
        out[name + "/sig_x0"] = np.asarray(m.sig_x0, float)
        out[name + "/sig_eta"] = np.asarray(m.sig_eta, float)
        out[name + "/zg"] = np.asarray(m.zg, float).reshape(-1)
        out[name + "/zg_term"] = np.asarray(m.zg_term, float).reshape(-1)
    save("models_vectors", out)


def _mpc_loop(policy, model, steps, out, rng, plant_noise=True):
    """Closed loop of scripts/mpc_state_est/mpc_quad.py:632-650 with a seeded NumPy plant."""
    x = np.array(model.x0, dtype=float).reshape(-1, 1)
    Lz = np.linalg.cholesky(model.sig_zeta)
    Le = np.linalg.cholesky(model.sig_eta)
    y = model.measure(x.T).T + Lz @ rng.normal(size=(Lz.shape[0], 1))
    u = np.zeros((model.dim_u, 1))
    ys, us_prev, mus, covs, ctrls, alphas = [], [], [], [], [], []
    for t in range(steps):
        ys.append(y[:, 0].copy())
        us_prev.append(u[:, 0].copy())
        u = policy(t, y, u)
        u = np.clip(u.T, model.xu_lim[0, model.dim_x:], model.xu_lim[1, model.dim_x:]).T
        mus.append(np.asarray(policy.mu, float).reshape(-1).copy())
        covs.append(np.asarray(policy.covar, float).copy())
        ctrls.append(u[:, 0].copy())
        alphas.append(float(policy.i2c.alpha))
        x = model.dynamics(np.hstack((x.T, u.T))).T
        if plant_noise:
            x = x + Le @ rng.normal(size=x.shape)
        y = model.measure(x.T).T + Lz @ rng.normal(size=(Lz.shape[0], 1))
    out["y"], out["u_prev"] = np.array(ys), np.array(us_prev)
    out["mu"], out["covar"], out["ctrl"], out["alpha_steps"] = np.array(mus), np.array(covs), np.array(ctrls), np.array(alphas)
    out["xu_plan_last"] = np.asarray(policy.xu_history[-1], float)[:, :, 0]


def case_mpc_pendulum(feedforward, name, quad=(1, 0, 0), rule=None, z_rows=None, hyper=None, **meta):
    """PartiallyObservedMpcPolicy (i2c/policy/mpc.py:113-182) on the pendulum with a stand-in
    measurement model y = observe_terminal(x) + N(0, sig_zeta) (the reference defines `measure`
    only for its Box2D quadrotor). Protocol of mpc_quad.py:624-650: calibrate_alpha, warm start,
    calibrate_alpha, then the closed loop with per-cell targets."""
    from i2c.policy.mpc import PartiallyObservedMpcPolicy

    hp = dict(H=10, steps=25, n_iter=2, warm=8, seed=5, Q=np.diag([1, 100.0, 1]), R=np.diag([2.0]), Qf=None, alpha=10.0, sig_u=2.0,
              sig_zeta=np.diag([1e-4, 1e-4, 1e-3]), amp=0.3)
    hp.update(hyper or {})
    H, steps, n_iter, warm = hp["H"], hp["steps"], hp["n_iter"], hp["warm"]
    rng = np.random.default_rng(hp["seed"])
    model = make_env_model("PendulumKnown", None)
    model.sig_zeta = hp["sig_zeta"]
    model.measure = lambda x: model.observe_terminal(x)
    model.dim_y = 3
    Q, R = hp["Q"], hp["R"]
    Qf = Q if hp["Qf"] is None else hp["Qf"]
    alpha0 = hp["alpha"]
    mu_u = 0.1 * rng.normal(size=(H, 1))
    sig_u = hp["sig_u"] * np.eye(1)
    z_rows = steps + H if z_rows is None else z_rows
    z_traj = np.tile(np.asarray(model.zg, float).reshape(1, -1), (z_rows, 1))
    z_traj[:, 2] = hp["amp"] * np.sin(np.linspace(0, 3, z_rows))  # a moving velocity target
    g = I2cGraph(model, H, Q, R, Qf, alpha0, 1.0, mu_u, sig_u, None, None, CubatureQuadrature(*quad) if rule is None else rule)
    g._propagate = True
    policy = PartiallyObservedMpcPolicy(g, n_iter, sig_u, np.copy(z_traj))
    policy.set_control(feedforward=feedforward)
    out = problem_inputs("PendulumKnown", model, H, Q, R, Qf, alpha0, 1.0, mu_u, sig_u, None, None, quad,
                         feedforward=bool(feedforward), steps=steps, n_iter=n_iter, warm=warm, **meta)
    out["z_traj"], out["sig_zeta"] = z_traj, model.sig_zeta
    g.calibrate_alpha()
    out["alpha_cal1"] = np.array(g.alpha)
    policy.optimize(warm, model.x0, model.sig_x0)
    g.calibrate_alpha()
    out["alpha_cal2"] = np.array(g.alpha)
    _mpc_loop(policy, model, steps, out, rng)
    save(name, out)


def case_mpc_pendulum_ff():
    case_mpc_pendulum(True, "mpc_pendulum_ff")


def case_mpc_pendulum_fb():
    case_mpc_pendulum(False, "mpc_pendulum_fb")


def case_mpc_pendulum_general_weights():
    """The graph infers with general cubature weights; the policy's state estimator keeps CubatureQuadrature(1, 0, 0)
    (mpc.py:121-123)."""
    case_mpc_pendulum(False, "mpc_pendulum_fb_general", quad=(1.2, 0.44, 0.5))


def case_mpc_pendulum_linearize():
    """MpcPolicy accepts any I2cGraph (mpc.py:16-33): the loop on a Linearize() graph (filter: unit cubature rule)."""
    from i2c.exp_types import Linearize

    case_mpc_pendulum(False, "mpc_pendulum_fb_lin", rule=Linearize(), inference="linearize", jacobian="complex-step stand-in (oracle/ref_shim.py)")


def case_mpc_pendulum_linearize_ff():
    from i2c.exp_types import Linearize

    case_mpc_pendulum(True, "mpc_pendulum_ff_lin", rule=Linearize(), inference="linearize", jacobian="complex-step stand-in (oracle/ref_shim.py)")


def case_mpc_pendulum_short_targets():
    """The target trajectory runs out during the loop: the appended cell then takes the target of the cell before it
    (mpc.py:75-78)."""
    case_mpc_pendulum(False, "mpc_pendulum_fb_short_targets", z_rows=22)


def case_mpc_pendulum_random():
    """Six MPC loops with random hyper-parameters (horizon, sweeps per step, cost weights incl. coupled ones, temperature, action
    prior, measurement noise, target amplitude) over the inference rules and both control modes."""
    from i2c.exp_types import Linearize

    rng = np.random.default_rng(2024)
    kinds = [("cubature", None, {}), ("general", None, {}), ("linearize", Linearize, dict(inference="linearize")),
             ("gh3", lambda: GaussHermiteQuadrature(3), dict(inference="gauss_hermite", gh_degree=3)),
             ("linearize", Linearize, dict(inference="linearize")), ("cubature", None, {})]
    for n, (kind, mk, meta) in enumerate(kinds):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        q = np.eye(3) + 0.15 * (q - np.eye(3))
        Q = q @ np.diag([1, 100.0, 1] * 10 ** rng.uniform(-0.4, 0.4, size=3)) @ q.T
        Q = 0.5 * (Q + Q.T)
        hyper = dict(H=int(rng.integers(4, 9)), steps=int(rng.integers(8, 14)), n_iter=int(rng.integers(1, 4)), warm=int(rng.integers(2, 6)),
                     seed=int(rng.integers(1 << 20)), Q=Q if n % 2 else np.diag(np.diag(Q)), R=np.diag([2.0 * 10 ** rng.uniform(-0.3, 0.3)]),
                     Qf=np.diag(np.diag(Q)) * 10 ** rng.uniform(-0.5, 0.5), alpha=float(10 ** rng.uniform(0.5, 1.5)),
                     sig_u=float(10 ** rng.uniform(-0.2, 0.5)), sig_zeta=np.diag(10.0 ** rng.uniform(-5, -3, size=3)),
                     amp=float(rng.uniform(0.1, 0.5)))
        case_mpc_pendulum(bool(n == 5), f"mpc_pendulum_random{n}", quad=(1.2, 0.44, 0.5) if kind == "general" else (1, 0, 0),
                          rule=None if mk is None else mk(), hyper=hyper, **meta)


def case_mpc_pendulum_gauss_hermite():
    case_mpc_pendulum(False, "mpc_pendulum_fb_gh3", rule=GaussHermiteQuadrature(3), inference="gauss_hermite", gh_degree=3)


def case_pendulum_tau(T=30, tau=7, n_detail=4, n_total=8):
    """A feedback horizon in the middle: _update_priors flips the cells with index <= tau (i2c.py:1210-1213), the rest stay
    feed-forward."""
    np.random.seed(9)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    g.tau = tau
    out = problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, (1, 0, 0), seed=9,
                         tau=tau)
    run_em(g, n_detail, n_total, out)
    out["feedforward_final"] = np.array([bool(c.state_action_independence) for c in g.cells])
    save(f"em_pendulum_T{T}_tau{tau}", out)


def _reference_quadrotor():
    """The build-defined analytic quadrotor (oracle/models_numpy.PlanarQuadrotor) wrapped in the
    REFERENCE's own model base classes, so that the reference solver and MPC policy run on it."""
    from i2c.env_def import BaseDef
    from i2c.model import BaseModelKnown
    from models_numpy import PlanarQuadrotor

    q = PlanarQuadrotor()

    class QuadrotorAnalytic(BaseDef, BaseModelKnown):
        name = "2D Quadrotor (analytic)"
        dim_x, dim_u, dim_z, dim_y = 6, 2, 8, 8
        dim_z_term = 6
        x0 = q.x0.reshape(-1, 1)
        sig_x0 = q.sig_x0
        sig_eta = q.sig_eta
        xag = q.zg_term.reshape(-1, 1)
        zg_term = xag
        sig_zeta = None
        gravity = q.gravity
        xu_lim = np.array([[-np.inf] * 6 + [0.0, 0.0], [np.inf] * 6 + [30.0, 30.0]])

        @staticmethod
        def dynamics(xu):
            return q.dynamics(xu)

        @staticmethod
        def observe(xu):
            return xu

        @staticmethod
        def observe_terminal(x):
            return x

        @staticmethod
        def measure(x):
            return q.measure(x)

    return QuadrotorAnalytic()


def case_em_quadrotor(T=20, n_detail=2, n_total=8):
    """Solver parity for config 4 with build-defined dynamics: reference I2cGraph on the analytic quadrotor."""
    model = _reference_quadrotor()
    rng = np.random.default_rng(2)
    Q = np.diag([1e3, 1e3, 1e3, 1, 1, 1]) / 1e3
    R = np.diag([1e-3, 1e-3])
    mu_u = 0.5 * model.gravity * np.ones((T, 2)) + 1e-2 * rng.normal(size=(T, 2))
    sig_u = 1e-2 * np.eye(2)
    g = I2cGraph(model, T, Q, R, Q, 1.0, 0.5, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("PlanarQuadrotor", model, T, Q, R, Q, 1.0, 0.5, mu_u, sig_u, None, None, (1, 0, 0))
    run_em(g, n_detail, n_total, out)
    save("em_quadrotor_T20", out)


def case_mpc_quadrotor(name="mpc_quadrotor_fb", H=10, steps=15, warm=10):
    """mpc_quad.py:538-650 (i2c, feedback, low noise) on the analytic quadrotor: tracking a moving target."""
    from i2c.policy.mpc import PartiallyObservedMpcPolicy

    n_iter = 2
    rng = np.random.default_rng(9)
    model = _reference_quadrotor()
    model.sig_zeta = np.diag([1e-6] * 8)
    W_, H_ = 20.0, 40.0 / 3.0
    z_traj = np.zeros((steps + H, 8))
    z_traj[:, 0] = np.linspace(W_ / 4, W_ / 4 + 2.0, steps + H)
    z_traj[:, 1] = H_ / 2 + 0.5 * np.sin(np.linspace(0, 2.0, steps + H))
    Q = np.diag([1e3, 1e3, 1e3, 1, 1, 1])
    R = np.diag([1e-3, 1e-3])
    Qf = Q / 1e3
    mu_u = 0.5 * model.gravity * np.ones((H, 2))
    sig_u = 1e-2 * np.eye(2)
    g = I2cGraph(model, H, Q, R, Qf, 1.0, 1.0, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    policy = PartiallyObservedMpcPolicy(g, n_iter, sig_u, np.copy(z_traj))
    policy.set_control(feedforward=False)
    out = problem_inputs("PlanarQuadrotor", model, H, Q, R, Qf, 1.0, 1.0, mu_u, sig_u, None, None, (1, 0, 0),
                         feedforward=False, steps=steps, n_iter=n_iter, warm=warm)
    out["z_traj"], out["sig_zeta"] = z_traj, model.sig_zeta
    g.calibrate_alpha()
    out["alpha_cal1"] = np.array(g.alpha)
    policy.optimize(warm, model.x0, model.sig_x0)
    g.calibrate_alpha()
    out["alpha_cal2"] = np.array(g.alpha)
    _mpc_loop(policy, model, steps, out, rng)
    save(name, out)


def _reference_quad12():
    """The build-defined 12-state quadrotor (oracle/models_numpy.Quadrotor12) wrapped in the REFERENCE's own model base
    classes, so that the reference solver and MPC policy run on it (as _reference_quadrotor does for the planar model)."""
    from i2c.env_def import BaseDef
    from i2c.model import BaseModelKnown
    from models_numpy import Quadrotor12

    q = Quadrotor12()

    class Quad12Analytic(BaseDef, BaseModelKnown):
        name = "3D Quadrotor (analytic)"
        dim_x, dim_u, dim_z, dim_y = 12, 4, 16, 9
        dim_z_term = 12
        x0 = q.x0.reshape(-1, 1)
        sig_x0 = q.sig_x0
        sig_eta = q.sig_eta
        xag = q.zg_term.reshape(-1, 1)
        zg_term = xag
        sig_zeta = None
        gravity = q.gravity
        xu_lim = np.array([[-np.inf] * 12 + [0.0] * 4, [np.inf] * 12 + [q.u_max] * 4])

        @staticmethod
        def dynamics(xu):
            return q.dynamics(xu)

        @staticmethod
        def observe(xu):
            return xu

        @staticmethod
        def observe_terminal(x):
            return x

        @staticmethod
        def measure(x):
            return q.measure(x)

        # Linearize() hooks (i2c.py:278-285, 321-323, 475-478): identity observations; the dynamics Jacobian through the
        # same stand-in for autograd.jacobian as the reference's own models get here (oracle/ref_shim.py)
        def dydxu(self, xu):
            import autograd

            return autograd.jacobian(self.dynamics)(xu).reshape((self.dim_x, self.dim_xu))

        def observe_linearize(self, xu):
            z = self.observe(xu).T
            return z, np.eye(16)[:, :12], np.zeros((16, 1)), np.eye(16)[:, 12:]

        def observe_terminal_linearize(self, x):
            return self.observe_terminal(x.T).T, np.eye(12), np.zeros((12, 1))

    return Quad12Analytic()


QUAD12_Q = np.diag([10.0] * 3 + [1.0] * 3 + [0.1] * 6)
QUAD12_R = 1e-2 * np.eye(4)


def case_em_quad12(T=20, n_detail=2, n_total=8):
    """Solver parity for config 4 at nx = 12: the reference I2cGraph on the build-defined 12-state quadrotor."""
    model = _reference_quad12()
    rng = np.random.default_rng(12)
    mu_u = 0.25 * model.gravity * np.ones((T, 4)) + 1e-2 * rng.normal(size=(T, 4))
    sig_u = 1e-2 * np.eye(4)
    g = I2cGraph(model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("Quadrotor12", model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, (1, 0, 0))
    run_em(g, n_detail, n_total, out)
    save("em_quad12_T20", out)


def case_em_quad12_propagate(T=12, n_detail=2, n_total=4):
    """The same with closed-loop propagation in every iteration (expert controller) and calibrate_alpha first."""
    model = _reference_quad12()
    rng = np.random.default_rng(13)
    mu_u = 0.25 * model.gravity * np.ones((T, 4)) + 1e-2 * rng.normal(size=(T, 4))
    sig_u = 1e-2 * np.eye(4)
    g = I2cGraph(model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = problem_inputs("Quadrotor12", model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, (1, 0, 0),
                         propagate=True)
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save("em_quad12_T12_propagate", out)


def case_em_quad12_covctrl(T=12, n_detail=3, n_total=6):
    """Covariance control (tempered terminal state prior, i2c.py:548-559) on the 12-state quadrotor, WITH a terminal cost (the
    terminal observation statistics then come from the pinned smoothed state, :565-570), propagation + KL every iteration."""
    model = _reference_quad12()
    rng = np.random.default_rng(21)
    mu_u = 0.25 * model.gravity * np.ones((T, 4)) + 1e-2 * rng.normal(size=(T, 4))
    sig_u = 1e-2 * np.eye(4)
    mu_xt = np.zeros(12)
    mu_xt[:3] = [0.05, -0.03, 0.08]
    A = rng.normal(size=(12, 12))
    sig_xt = 1e-3 * (np.eye(12) + 0.08 * (A + A.T) / np.abs(A).max())  # coupled, positive definite
    assert np.all(np.linalg.eigvalsh(sig_xt) > 0)
    g = I2cGraph(model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, mu_xt, sig_xt, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = problem_inputs("Quadrotor12", model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, mu_xt, sig_xt, (1, 0, 0),
                         propagate=True)
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save("em_quad12_covctrl_T12", out)


def _coupled(W, rng, strength=0.3):
    """A symmetric positive-definite weight with off-diagonal entries: D^1/2 (I + s (A + A^T) / (2 |A|)) D^1/2."""
    n = W.shape[0]
    A = rng.normal(size=(n, n))
    A = 0.5 * (A + A.T)
    np.fill_diagonal(A, 0.0)
    A *= strength / np.abs(np.linalg.eigvalsh(A)).max()
    d = np.sqrt(np.diag(W))
    return (np.eye(n) + A) * d[:, None] * d[None, :]


def case_em_quad12_nondiag(T=12, n_detail=2, n_total=5):
    """NON-DIAGONAL Q, R and Qf (i2c.py:781-789 takes any symmetric weight) on the 12-state quadrotor, with closed-loop
    propagation (its cost statistics use the same weights): pins the general-weight cost of the multi-lane kernels."""
    model = _reference_quad12()
    rng = np.random.default_rng(15)
    Q, R, Qf = _coupled(QUAD12_Q, rng), _coupled(QUAD12_R, rng), _coupled(QUAD12_Q / 5.0, rng)
    mu_u = 0.25 * model.gravity * np.ones((T, 4)) + 1e-2 * rng.normal(size=(T, 4))
    sig_u = 1e-2 * np.eye(4)
    g = I2cGraph(model, T, Q, R, Qf, 1.0, 0.5, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    g._propagate = True
    out = problem_inputs("Quadrotor12", model, T, Q, R, Qf, 1.0, 0.5, mu_u, sig_u, None, None, (1, 0, 0), propagate=True)
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save("em_quad12_nondiag_T12", out)


def case_em_dcp_nondiag(T=30, n_detail=2, n_total=6):
    """Non-diagonal weights on the double cartpole (nz = 9 with sin / cos features): the general-weight cost with a
    non-identity observation, for the lane and the group kernels."""
    model = make_env_model("DoubleCartpoleKnown", None)
    rng = np.random.default_rng(16)
    sf = 1e-3
    Q = _coupled(sf * np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0]), rng)
    R = sf * np.diag([0.1])
    Qf = _coupled(sf * np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0]), rng)
    mu_u = 1e-2 * rng.normal(size=(T, 1))
    g = I2cGraph(model, T, Q, R, Qf, 0.05, 0.99, mu_u, np.eye(1), None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("DoubleCartpoleKnown", model, T, Q, R, Qf, 0.05, 0.99, mu_u, np.eye(1), None, None, (1, 0, 0))
    run_em(g, n_detail, n_total, out)
    save("em_dcp_nondiag_T30", out)


def case_mpc_quad12(rule=None, name="mpc_quad12_fb", H=8, steps=10, warm=6, **meta):
    """mpc_quad.py:538-650 (i2c, feedback, low noise) on the 12-state quadrotor: tracking a moving position target."""
    from i2c.policy.mpc import PartiallyObservedMpcPolicy

    n_iter = 2
    rng = np.random.default_rng(14)
    model = _reference_quad12()
    model.sig_zeta = 1e-6 * np.eye(9)
    z_traj = np.zeros((steps + H, 16))
    z_traj[:, 0] = np.linspace(0.0, 0.6, steps + H)
    z_traj[:, 1] = 0.2 * np.sin(np.linspace(0, 2.0, steps + H))
    z_traj[:, 2] = np.linspace(0.0, 0.3, steps + H)
    z_traj[:, 12:] = 0.25 * model.gravity
    Q, R = QUAD12_Q, QUAD12_R
    Qf = Q / 10.0
    mu_u = 0.25 * model.gravity * np.ones((H, 4))
    sig_u = 1e-2 * np.eye(4)
    g = I2cGraph(model, H, Q, R, Qf, 1.0, 1.0, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0) if rule is None else rule)
    g._propagate = True
    policy = PartiallyObservedMpcPolicy(g, n_iter, sig_u, np.copy(z_traj))
    policy.set_control(feedforward=False)
    out = problem_inputs("Quadrotor12", model, H, Q, R, Qf, 1.0, 1.0, mu_u, sig_u, None, None, (1, 0, 0),
                         feedforward=False, steps=steps, n_iter=n_iter, warm=warm, **meta)
    out["z_traj"], out["sig_zeta"] = z_traj, model.sig_zeta
    g.calibrate_alpha()
    out["alpha_cal1"] = np.array(g.alpha)
    policy.optimize(warm, model.x0, model.sig_x0)
    g.calibrate_alpha()
    out["alpha_cal2"] = np.array(g.alpha)
    _mpc_loop(policy, model, steps, out, rng)
    save(name, out)


def case_mpc_quad12_H50():
    """BASELINE config 4's full horizon (H = 50, mpc_iter = 2 as in mpc_quad.py:559) through the reference's own
    PartiallyObservedMpcPolicy: four control steps after a four-iteration warm start (minutes of reference time)."""
    case_mpc_quad12(None, "mpc_quad12_fb_H50", H=50, steps=4, warm=4)


def case_mpc_quadrotor_H50():
    """The planar quadrotor (the reference's actual config-4 model class, mpc_quad.py:219-383) at the full horizon."""
    case_mpc_quadrotor(name="mpc_quadrotor_fb_H50", H=50, steps=4, warm=4)


def case_mpc_quad12_linearize():
    """The same loop on a Linearize() graph: the wave kernels' Linearize variant on the ring, the terminal cost at the end of the
    chain with the appended cell's own temperature."""
    from i2c.exp_types import Linearize

    case_mpc_quad12(Linearize(), "mpc_quad12_fb_lin", inference="linearize", jacobian="complex-step stand-in (oracle/ref_shim.py)")


def case_rollouts(T=60, n_em=4):
    """Row f3: the reference's rollout / evaluation path on its own PendulumKnown simulator --
    BaseSim.run (i2c/env.py:40-74) driven by TimeIndexedLinearGaussianPolicy / ExpertTimeIndexedLinearGaussianPolicy
    (i2c/policy/linear.py:8-100) filled from an I2cGraph, and StochasticTrajectoryEvaluator.eval (i2c/utils.py:150-192).
    Deterministic plant + deterministic policy runs are RNG-free. The stochastic run records every sample the reference draws
    (numpy.random.multivariate_normal inside env.forward and the policy) and stores it STANDARDISED (eps = chol(cov)^-1
    (sample - mean)), which is the form i2c_rollout takes its disturbances in."""
    import i2c.env as ref_env
    import i2c.policy.linear as ref_lin
    from i2c.utils import StochasticTrajectoryEvaluator

    rng = np.random.default_rng(21)
    model = make_env_model("PendulumKnown", None)
    Q, R = np.diag([1.0, 100.0, 1.0]), np.diag([2.0])
    mu_u = 1e-2 * rng.normal(size=(T, 1))
    sig_u = 2.0 * np.eye(1)
    g = I2cGraph(model, T, Q, R, Q, 100.0, 0.0, mu_u, sig_u, None, None, CubatureQuadrature(1, 0, 0))
    out = problem_inputs("PendulumKnown", model, T, Q, R, Q, 100.0, 0.0, mu_u, sig_u, None, None, (1, 0, 0), n_em=n_em)
    for _ in range(n_em):
        g.learn_msgs()
    K, k, sigK = g.get_local_linear_policy()
    Ke, ke, sKe, mue, lame = g.get_local_expert_linear_policy()
    out.update({"K": K, "k": k, "sigK": sigK, "expert/K": Ke, "expert/k": ke, "expert/sigK": sKe, "expert/mu": mue, "expert/lam": lame})
    env = ref_env.PendulumKnown(T)
    lin = ref_lin.TimeIndexedLinearGaussianPolicy(sig_u, T, 1, 2)
    lin.write(K, k, sigK)
    pols = {"linear": lin}
    for soft in (True, False):
        pe = ref_lin.ExpertTimeIndexedLinearGaussianPolicy(sig_u, T, 1, 2, soft=soft)
        pe.write(Ke, ke, sKe, mue, lame)
        pols["expert_soft" if soft else "expert_hard"] = pe

    def put(tag, res):
        xt, yt, zt, z_term = res
        out[tag + "/xu"], out[tag + "/dx"], out[tag + "/z"], out[tag + "/z_term"] = xt, yt, zt, np.asarray(z_term, float).reshape(-1)

    env.deterministic = True
    for name, pol in pols.items():
        put("det/" + name, env.run(pol, deterministic=True))

    # stochastic plant and stochastic policy, every draw recorded
    draws = []
    real_mvn = np.random.multivariate_normal

    def recording_mvn(mean, cov, size=None, **kw):
        smp = real_mvn(mean, cov, size, **kw)
        draws.append((np.asarray(mean, float).reshape(-1), np.asarray(cov, float), np.asarray(smp, float).reshape(-1)))
        return smp

    env.deterministic = False
    ref_env.mvn = recording_mvn
    ref_lin.mvn = recording_mvn
    z_runs, zt_runs = [], []
    try:
        for name in ("linear", "expert_soft"):
            eps_x, eps_u = np.zeros((3, T, 2)), np.zeros((3, T, 1))
            for rr in range(3):
                np.random.seed(100 + rr)
                del draws[:]
                res = env.run(pols[name], deterministic=False)
                put(f"sto/{name}/{rr}", res)
                assert len(draws) == 2 * T  # per step: the policy's action sample, then the plant's disturbance
                for t in range(T):
                    mu_a, cov_a, smp_a = draws[2 * t]
                    mu_p, cov_p, smp_p = draws[2 * t + 1]
                    eps_u[rr, t] = np.linalg.solve(np.linalg.cholesky(cov_a), smp_a - mu_a)
                    eps_x[rr, t] = np.linalg.solve(np.linalg.cholesky(cov_p), smp_p - mu_p)
                if name == "linear":
                    z_runs.append(res[2])
                    zt_runs.append(res[3])
            out[f"sto/{name}/eps_x"], out[f"sto/{name}/eps_u"] = eps_x, eps_u
    finally:
        ref_env.mvn = real_mvn
        ref_lin.mvn = real_mvn
        env.deterministic = False

    # the evaluator of scripts/i2c_run.py:66-75, 104-106 on those rollouts against the plan
    z_est, z_term_est = g.get_marginal_observed_trajectory()
    ev = StochasticTrajectoryEvaluator(g.QR, g.Qf, g.z, g.z_term, g.Qf.shape[0])
    ev.eval(z_runs, zt_runs, z_est, z_term_est)
    ev.eval(z_runs[:2], zt_runs[:2], z_est, z_term_est)
    out["eval/z_est"], out["eval/z_term_est"] = np.asarray(z_est, float), np.asarray(z_term_est, float)
    for key in ("mu_actual_cost", "min_actual_cost", "max_actual_cost", "actual_cost_10", "actual_cost_90", "planned_cost"):
        out["eval/" + key] = np.asarray(getattr(ev, key), float).reshape(-1)
    save("rollouts_pendulum_T60", out)


def case_i2c_run(config="pendulum_known_quad", name="run_pendulum_seed0"):
    """The reference's own runner, scripts/i2c_run.py:run(), on its shipped pendulum config (seed 0,
    N_INFERENCE cut to 6): what a user sees -- costs_m, alphas, the saved plan and the final policy."""
    import importlib
    import tempfile

    np.random.seed(0)  # i2c_run.py:215 set_seed(args.random_seed) happens BEFORE the config import draws mu_u
    runner = importlib.import_module("i2c_run")
    experiment = importlib.import_module("experiments." + config)
    experiment.N_INFERENCE = 6
    experiment.N_ITERS_PER_PLOT = 100
    captured = {}
    real_graph = runner.I2cGraph

    def capture(*a, **k):
        captured["i2c"] = real_graph(*a, **k)
        return captured["i2c"]

    runner.I2cGraph = capture
    with tempfile.TemporaryDirectory() as res_dir:
        runner.run(experiment, res_dir, None)
        out = {k: np.load(os.path.join(res_dir, k + ".npy")) for k in ("xu_plan", "x_plan", "u_plan", "z_plan",
                                                                         "xu_real", "dx_real", "x_real", "u_real")}
    runner.I2cGraph = real_graph
    g = captured["i2c"]
    out["mu_u"] = np.asarray(experiment.INFERENCE.mu_u, float)
    out["costs_m"] = np.asarray(g.costs_m, float)
    out["alphas"] = np.asarray(g.alphas, float)
    out["alphas_desired"] = np.asarray(g.alphas_desired, float)
    K, k, sigK = g.get_local_linear_policy()
    out["K"], out["k"], out["sigK"] = K, k, sigK
    out["meta"] = np.array(json.dumps(dict(config=config, seed=0, n_inference=6, T=int(experiment.N_DURATION))))
    save(name, out)


def case_i2c_run_linearize():
    """The same runner on the shipped Linearize config scripts/experiments/pendulum_known.py."""
    case_i2c_run("pendulum_known", "run_pendulum_linearize_seed0")


# --------------------------------------------------------------------------------------
# Linearize inference (i2c.py:244-348, 449-542, 612-678). Linear systems need no Jacobian (model.py:227-229): those
# cases pin the solver algebra exactly. The nonlinear ones use ref_shim's complex-step stand-in for autograd.jacobian.
LIN_KEYS_F = FWD_KEYS + ["A", "B", "a", "E", "F"]


def _capture_lin(g, out, it, fwd):
    keys = LIN_KEYS_F if fwd else BWD_KEYS
    for k in keys:
        arrs = [np.asarray(getattr(c, k), dtype=float) for c in g.cells]
        a = np.stack(arrs)
        if k.startswith("mu_") or k in ("k", "a"):
            a = a.reshape(a.shape[0], -1)
        out[f"it{it}/{k}"] = a


def run_em_linearize(g, n_detail, n_total, out, pre_propagate=False):
    from i2c.exp_types import Linearize  # noqa: F401

    if pre_propagate:
        g.propagate()
    for it in range(1, n_total + 1):
        g.em_iter += 1
        g._forward_msgs()
        if it <= n_detail:
            _capture_lin(g, out, it, True)
        g._backward_msgs()
        if it <= n_detail:
            _capture_lin(g, out, it, False)
            c = g.cells[-1]
            out[f"it{it}/mu_z3_m"] = np.asarray(c.mu_z3_m, dtype=float).reshape(-1)
            out[f"it{it}/sig_z3_m"] = np.asarray(c.sig_z3_m, dtype=float)
        if g._propagate:
            g.propagate()
            if it <= n_detail:
                for k, v in capture(g, PF_KEYS).items():
                    out[f"it{it}/{k}"] = v
        g._maximize()
    out["alphas"] = np.asarray(g.alphas, dtype=float)
    out["alphas_desired"] = np.asarray(g.alphas_desired, dtype=float)
    out["costs_m"] = np.asarray(g.costs_m, dtype=float)
    out["costs_m_var"] = np.asarray(g.costs_m_var, dtype=float)
    out["costs_pf"] = np.asarray(g.costs_pf, dtype=float)
    if g._propagate:
        out["alphas_pf"] = np.asarray(g.alphas_pf, dtype=float)
        out["costs_pf_var"] = np.asarray(g.costs_pf_var, dtype=float)
    if len(g.kl_terms):
        out["kl_terms"] = np.asarray(g.kl_terms, dtype=float)
    K, k, sigK = g.get_local_linear_policy()
    out["final/K"], out["final/k"], out["final/sigK"] = K, k, sigK
    mu, sig = g.get_marginal_state_action_distribution()
    out["final/mu_xu0_m"], out["final/sig_xu0_m"] = mu, sig


def case_lin_linear(T=60, n_detail=2, n_total=10, noise=1e-4):
    """scripts/experiments/linear_known.py:20-31 with non-degenerate noise (the shipped 1e-20 makes sig_x3_f
    numerically singular: cond ~1e20, no digits are reproducible)."""
    from i2c.exp_types import Linearize

    mu_u = np.zeros((T, 1))
    Q, R = np.diag([10.0, 10.0]), np.diag([1.0])
    model = make_env_model("LinearKnown", None)
    model.sig_x0 = noise * np.eye(2)
    model.sig_eta = noise * np.eye(2)
    g = I2cGraph(model, T, Q, R, Q, 1e2, 0.0, mu_u, 1e2 * np.eye(1), None, None, Linearize())
    out = problem_inputs("LinearKnown", model, T, Q, R, Q, 1e2, 0.0, mu_u, 1e2 * np.eye(1), None, None, (1, 0, 0),
                         noise=noise, inference="linearize")
    run_em_linearize(g, n_detail, n_total, out)
    save("lin_linear_T60", out)


def case_lin_lqr_compare(noise=1e-20):  # the shipped value: this path has no cancellation problem with it
    """The protocol of scripts/lqr_compare.py:120-175 (config 0): redefined linear system, alpha 1e-5, feed-forward
    cells, one forward/backward pass, then the Riccati messages; LQR solution from utils.finite_horizon_lqr."""
    from i2c.exp_types import Linearize
    from i2c.utils import finite_horizon_lqr
    import experiments.linear_known as experiment

    T = experiment.N_DURATION
    model = make_env_model("LinearKnown", None)
    model.sig_x0 = noise * np.eye(2)
    model.sig_eta = noise * np.eye(2)
    model.xag = 10 * np.ones((2, 1))
    model.zg_term = 10 * np.ones((2, 1))
    model.a = model.xag - model.A @ model.xag
    Q, R, Qf = experiment.INFERENCE.Q, experiment.INFERENCE.R, experiment.INFERENCE.Qf
    x_lqr, u_lqr, K_lqr, k_lqr, cost_lqr, P, p = finite_horizon_lqr(
        T, model.A, model.a[:, 0], model.B, Q, R, model.x0[:, 0], model.xag[:, 0], np.zeros((1,)), 2, 1)
    mu_u = np.zeros((T, 1))
    g = I2cGraph(model, T, Q, R, Qf, 1e-5, experiment.INFERENCE.alpha_update_tol, mu_u, 1e2 * np.eye(1), None, None, Linearize())
    g.use_expert_controller = False
    out = problem_inputs("LinearKnown", model, T, Q, R, Qf, 1e-5, experiment.INFERENCE.alpha_update_tol, mu_u, 1e2 * np.eye(1),
                         None, None, (1, 0, 0), noise=noise, inference="linearize", goal=10.0, use_expert_controller=False)
    g._forward_backward_msgs()
    _capture_lin(g, out, 1, True)
    _capture_lin(g, out, 1, False)
    g._backward_ricatti_msgs()
    for k in ("K", "k", "sigK", "nu_x0_b", "lambda_x0_b"):
        a = np.stack([np.asarray(getattr(c, k), float) for c in g.cells])
        out["riccati/" + k] = a.reshape(a.shape[0], -1) if k in ("k", "nu_x0_b") else a
    out["lqr/x"], out["lqr/u"], out["lqr/K"], out["lqr/k"] = x_lqr, u_lqr, K_lqr, k_lqr
    out["lqr/P"], out["lqr/p"] = np.asarray(P, float), np.asarray(p, float)
    save("lin_lqr_compare", out)


def case_lin_covariance_control(n_detail=2):
    """scripts/linear_gaussian_covariance_control.py:97-126 + experiments/linear_known_covariance_control.py."""
    import experiments.linear_known_covariance_control as experiment

    T, inf = experiment.N_DURATION, experiment.INFERENCE
    model = make_env_model(experiment.ENVIRONMENT, None)
    g = I2cGraph(model, T, inf.Q, inf.R, inf.Qf, inf.alpha, inf.alpha_update_tol, inf.mu_u, inf.sig_u, inf.mu_x_term,
                 inf.sig_x_term, inf.inference)
    for c in g.cells:
        c.use_expert_controller = False
    g._propagate = True
    out = problem_inputs(experiment.ENVIRONMENT, model, T, inf.Q, inf.R, inf.Qf, inf.alpha, inf.alpha_update_tol, inf.mu_u,
                         inf.sig_u, inf.mu_x_term, inf.sig_x_term, (1, 0, 0), inference="linearize", propagate=True,
                         use_expert_controller=False)
    run_em_linearize(g, n_detail, experiment.N_INFERENCE, out)
    save("lin_covctrl_T50", out)


def case_lin_covctrl_terminal_cost(T=30, n_detail=2, n_total=6, noise=1e-4):
    """Linearize() with covariance control AND a terminal cost: the one combination in which the back-calculated
    sig_xi_terminal of i2c.py:455-462 reaches a result (sig_z3_m :499-501 -> alpha :989-992). The terminal covariance is
    tighter than the filtered one in one direction and looser in the other, so the multiplier is indefinite."""
    from i2c.exp_types import Linearize

    mu_u = np.zeros((T, 1))
    Q, R = np.diag([10.0, 10.0]), np.diag([1.0])
    mu_T, sig_T = np.array([[1.0], [0.5]]), np.array([[1e-3, 2e-4], [2e-4, 50.0]])
    model = make_env_model("LinearKnown", None)
    model.sig_x0 = noise * np.eye(2)
    model.sig_eta = noise * np.eye(2)
    g = I2cGraph(model, T, Q, R, Q, 1e2, 0.0, mu_u, 1e2 * np.eye(1), mu_T, sig_T, Linearize())
    out = problem_inputs("LinearKnown", model, T, Q, R, Q, 1e2, 0.0, mu_u, 1e2 * np.eye(1), mu_T, sig_T, (1, 0, 0),
                         noise=noise, inference="linearize")
    run_em_linearize(g, n_detail, n_total, out)
    save("lin_covctrl_qf_T30", out)


def _case_lin_nonlinear(env, T, Q, R, Qf, alpha, tol, mu_u, sig_u, n_detail, n_total, name):
    from i2c.exp_types import Linearize

    model = make_env_model(env, None)
    g = I2cGraph(model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, None, None, Linearize())
    out = problem_inputs(env, model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, None, None, (1, 0, 0), inference="linearize",
                         jacobian="complex-step stand-in for autograd.jacobian (oracle/ref_shim.py)")
    run_em_linearize(g, n_detail, n_total, out)
    save(name, out)


def case_lin_pendulum():
    """scripts/experiments/pendulum_known.py:21-33, but with a small random initial action sequence: with the shipped
    mu_u = 0 the pendulum hangs at rest and the first iterations only move rounding noise."""
    T = 100
    np.random.seed(2)
    _case_lin_nonlinear("PendulumKnown", T, np.diag([1, 100.0, 1]), np.diag([1.0]), np.diag([1, 100.0, 1]), 100.0, 0.99,
                        1e-1 * np.random.randn(T, 1), 0.2 * np.eye(1), 2, 40, "lin_pendulum_T100")


def case_lin_cartpole():
    """scripts/experiments/cartpole_known.py:21-33 at a shorter horizon."""
    T = 100
    np.random.seed(0)
    Q = np.diag([1.0, 1.0, 100.0, 10.0, 1.0])
    _case_lin_nonlinear("CartpoleKnown", T, Q, np.diag([1.0]), Q, 70.0, 0.99, 1e-2 * np.random.randn(T, 1), 0.25 * np.eye(1),
                        2, 10, "lin_cartpole_T100")


def case_lin_double_cartpole():
    """scripts/experiments/double_cartpole_known.py:20-32 at a shorter horizon, random initial actions (see above)."""
    T = 80
    np.random.seed(3)
    Q = np.diag([1.0, 1.0, 100.0, 1.0, 100.0, 1.0, 1.0, 1.0])
    Qf = np.diag([1.0, 1000.0, 1000.0, 1000.0, 1000.0, 100.0, 100.0, 100.0])
    _case_lin_nonlinear("DoubleCartpoleKnown", T, Q, np.diag([0.1]), Qf, 90.0, 0.9995, 1e-1 * np.random.randn(T, 1),
                        0.04 * np.eye(1), 2, 8, "lin_dcp_T80")



def case_lin_quad12(T=20, n_detail=2, n_total=6):
    """Linearize() on the build-defined 12-state quadrotor (d = 16) through the reference's own I2cGraph."""
    from i2c.exp_types import Linearize

    model = _reference_quad12()
    rng = np.random.default_rng(17)
    mu_u = 0.25 * model.gravity * np.ones((T, 4)) + 1e-2 * rng.normal(size=(T, 4))
    sig_u = 1e-2 * np.eye(4)
    g = I2cGraph(model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, Linearize())
    out = problem_inputs("Quadrotor12", model, T, QUAD12_Q, QUAD12_R, QUAD12_Q, 1.0, 0.5, mu_u, sig_u, None, None, (1, 0, 0),
                         inference="linearize", jacobian="complex-step stand-in for autograd.jacobian (oracle/ref_shim.py)")
    run_em_linearize(g, n_detail, n_total, out)
    save("lin_quad12_T20", out)


def case_gh_pendulum(T=40, degree=3, n_detail=2, n_total=8):
    """GaussHermiteQuadrature(degree) as the inference method (exp_types.py:52-68): 27 points per joint transform, 9 for
    the terminal one. No reference script ships such a config; hyper-parameters of pendulum_known_quad.py."""
    np.random.seed(4)
    mu_u = 1e-2 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([2.0]), np.diag([1, 100.0, 1])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, GaussHermiteQuadrature(degree))
    g._propagate = True
    out = problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.0, mu_u, 2.0 * np.eye(1), None, None, (1, 0, 0),
                         seed=4, inference="gauss_hermite", gh_degree=degree, propagate=True)
    run_em(g, n_detail, n_total, out)
    save(f"gh{degree}_pendulum_T{T}", out)


def case_gh_covariance_control(T=100, degree=3, n_detail=2, n_total=30):
    """Covariance control (tempered terminal prior + KL term) under GaussHermiteQuadrature(3), propagation with the grid.
    (A small random initial action sequence: with mu_u = 0 the observed mean is rounding noise around zero.)"""
    np.random.seed(8)
    mu_u = 1e-2 * np.random.randn(T, 1)
    R = np.diag([1.0])
    mu_xt, sig_xt = np.array([0.0, 0.0]), np.diag([1e-3, 1e-3])
    model = make_env_model("PendulumKnownActReg", None)
    g = I2cGraph(model, T, None, R, None, 300.0, 1.0, mu_u, 0.5 * np.eye(1), mu_xt, sig_xt, GaussHermiteQuadrature(degree))
    for c in g.cells:
        c.use_expert_controller = False
    g._propagate = True
    out = problem_inputs("PendulumKnownActReg", model, T, None, R, None, 300.0, 1.0, mu_u, 0.5 * np.eye(1), mu_xt, sig_xt, (1, 0, 0),
                         propagate=True, use_expert_controller=False, inference="gauss_hermite", gh_degree=degree)
    run_em(g, n_detail, n_total, out, pre_propagate=True)
    save(f"gh{degree}_covctrl_T{T}", out)


def case_lin_pendulum_propagate(T=40, n_detail=2, n_total=8):
    """Linearize() on the pendulum WITH closed-loop propagation (unit cubature rule, i2c.py:109-115) and the expert
    controller in both the forward messages (:259-265) and the propagation (:160-167)."""
    from i2c.exp_types import Linearize

    np.random.seed(6)
    mu_u = 1e-1 * np.random.randn(T, 1)
    Q, R, Qf = np.diag([1, 100.0, 1]), np.diag([1.0]), np.diag([1, 100.0, 1])
    model = make_env_model("PendulumKnown", None)
    g = I2cGraph(model, T, Q, R, Qf, 100.0, 0.99, mu_u, 0.2 * np.eye(1), None, None, Linearize())
    g._propagate = True
    out = problem_inputs("PendulumKnown", model, T, Q, R, Qf, 100.0, 0.99, mu_u, 0.2 * np.eye(1), None, None, (1, 0, 0), seed=6,
                         inference="linearize", propagate=True, use_expert_controller=True,
                         jacobian="complex-step stand-in for autograd.jacobian (oracle/ref_shim.py)")
    run_em_linearize(g, n_detail, n_total, out, pre_propagate=True)
    save("lin_pendulum_T40_propagate", out)


def case_gh_linear(T=30, degree=4, n_detail=2, n_total=6, noise=1e-4):
    """Gauss-Hermite degree 4 on LinearKnown (non-degenerate noise): 64 points per joint transform."""
    mu_u = np.zeros((T, 1))
    Q, R = np.diag([10.0, 10.0]), np.diag([1.0])
    model = make_env_model("LinearKnown", None)
    model.sig_x0 = noise * np.eye(2)
    model.sig_eta = noise * np.eye(2)
    g = I2cGraph(model, T, Q, R, Q, 800.0, 0.0, mu_u, np.eye(1), None, None, GaussHermiteQuadrature(degree))
    out = problem_inputs("LinearKnown", model, T, Q, R, Q, 800.0, 0.0, mu_u, np.eye(1), None, None, (1, 0, 0), noise=noise,
                         inference="gauss_hermite", gh_degree=degree)
    run_em(g, n_detail, n_total, out)
    save(f"gh{degree}_linear_T{T}", out)


CASES = {
    "pendulum": case_pendulum,
    "pendulum_long": case_pendulum_long,
    "pendulum_seeds": case_pendulum_seeds,
    "pendulum_general": case_pendulum_general_weights,
    "dcp": case_double_cartpole,
    "dcp300": case_double_cartpole_T300,
    "dcp300_long": case_double_cartpole_T300_long,
    "cartpole": case_cartpole,
    "linear": case_linear,
    "covctrl": case_covariance_control,
    "covctrl_qf": case_covariance_control_terminal_cost,
    "propagate": case_propagate_expert,
    "quadrature": case_quadrature,
    "models": case_models,
    "mpc_ff": case_mpc_pendulum_ff,
    "mpc_fb": case_mpc_pendulum_fb,
    "mpc_fb_general": case_mpc_pendulum_general_weights,
    "pendulum_tau": case_pendulum_tau,
    "mpc_fb_lin": case_mpc_pendulum_linearize,
    "mpc_fb_gh": case_mpc_pendulum_gauss_hermite,
    "mpc_fb_short": case_mpc_pendulum_short_targets,
    "mpc_random": case_mpc_pendulum_random,
    "mpc_ff_lin": case_mpc_pendulum_linearize_ff,
    "em_quad": case_em_quadrotor,
    "mpc_quad": case_mpc_quadrotor,
    "em_quad12": case_em_quad12,
    "em_quad12_pf": case_em_quad12_propagate,
    "mpc_quad12": case_mpc_quad12,
    "mpc_quad12_lin": case_mpc_quad12_linearize,
    "mpc_quad12_H50": case_mpc_quad12_H50,
    "mpc_quad_H50": case_mpc_quadrotor_H50,
    "em_quad12_nondiag": case_em_quad12_nondiag,
    "em_quad12_covctrl": case_em_quad12_covctrl,
    "em_dcp_nondiag": case_em_dcp_nondiag,
    "rollouts": case_rollouts,
    "i2c_run": case_i2c_run,
    "i2c_run_lin": case_i2c_run_linearize,
    "gh_pendulum": case_gh_pendulum,
    "gh_linear": case_gh_linear,
    "gh_covctrl": case_gh_covariance_control,
    "lin_pendulum_pf": case_lin_pendulum_propagate,
    "lin_linear": case_lin_linear,
    "lin_lqr": case_lin_lqr_compare,
    "lin_covctrl": case_lin_covariance_control,
    "lin_covctrl_qf": case_lin_covctrl_terminal_cost,
    "lin_pendulum": case_lin_pendulum,
    "lin_cartpole": case_lin_cartpole,
    "lin_dcp": case_lin_double_cartpole,
    "lin_quad12": case_lin_quad12,
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    for c in which:
        CASES[c]()
