"""A second out-of-tree model (tests/plugins/spring_chain.hpp): three states, one action, the IDENTITY observation of (x, u) -- the
model class whose cost update works on the Cholesky factor of the joint prior in the quad forward kernel (q_kalman_sqrt), here
with the whole joint in ONE 4 x 4 block that the action shares with three state rows (the in-tree models with that update have the
actions in a block of their own, or two and two). Built from its header, registered at run time (test_model_plugin.py has the ABI
checks), every kernel family against the CPU oracle fed the same model in NumPy -- cubature EM incl. a terminal cost, Linearize,
closed-loop propagation. CPU: host simulation; `-m gpu`: the hipcc build on cuda:0."""
import importlib
import os

import numpy as np
import pytest

from parity import close, np_

pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import KnownModel, make_env_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "tests", "plugins", "spring_chain.hpp")
DT, K, H, C_, A, U_MAX = 0.05, 2.0, 4.0, 0.3, 1.5, 2.0


class SpringChainNumpy:
    """NumPy twin of tests/plugins/spring_chain.hpp in the oracle's protocol (oracle/models_numpy.py): vectorised over leading axes."""

    name = "SpringChain"
    dim_x, dim_u, dim_z, dim_z_term = 3, 1, 4, 3
    has_terminal_obs = True

    def __init__(self):
        self.x0 = np.array([0.8, 0.0, 0.0])
        self.sig_x0 = 1e-4 * np.eye(3)
        self.sig_eta = 1e-5 * np.eye(3)
        self.zg = np.zeros(4)
        self.zg_term = np.zeros(3)

    @property
    def dim_xu(self):
        return 4

    def dynamics(self, xu):
        u = np.clip(xu[..., 3], -U_MAX, U_MAX)
        w = xu[..., 2] + DT * (u - A * xu[..., 2])
        v = xu[..., 1] + DT * (xu[..., 2] - K * xu[..., 0] - H * xu[..., 0] ** 3 - C_ * xu[..., 1])
        return np.stack((xu[..., 0] + DT * v, v, w), axis=-1)

    def observe(self, xu):
        return np.array(xu[..., :4])  # (no dtype: the Linearize oracle differentiates by complex steps)

    def observe_terminal(self, x):
        return np.array(x[..., :3])


class SpringChainKnown(KnownModel):
    name = "SpringChain"
    model_name = "SpringChainKnown"
    hip_header = HEADER  # struct SpringChain, library spring_chain: derived from the file name
    dim_x, dim_u, dim_z, dim_z_term = 3, 1, 4, 3

    def __init__(self):
        super().__init__()
        n = SpringChainNumpy()
        self._np = n
        self.x0 = n.x0.reshape(3, 1)
        self.xag = np.zeros((4, 1))
        self.sig_x0, self.sig_eta = n.sig_x0, n.sig_eta
        self.xu_lim = np.array([[-np.inf, -np.inf, -np.inf, -U_MAX], [np.inf, np.inf, np.inf, U_MAX]])

    @property
    def zg(self):
        return np.zeros((4, 1))

    @property
    def zg_term(self):
        return np.zeros((3, 1))

    def device_params(self):
        return [DT, K, H, C_, A, U_MAX]

    def dynamics(self, xu):
        return self._np.dynamics(np.asarray(xu, float))

    def observe(self, xu):
        return self._np.observe(np.asarray(xu, float))

    def observe_terminal(self, x):
        return self._np.observe_terminal(np.asarray(x, float))


def problem(B=6, T=30, seed=5):
    rng = np.random.default_rng(seed)
    x0 = np.array([0.8, 0.0, 0.0]) + 5e-2 * rng.normal(size=(B, 3))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    # a full (non-diagonal) state weight: the matrix branch of the update's N^-1
    Lq = np.eye(3) + 0.2 * np.tril(rng.normal(size=(3, 3)), -1)
    Q, R, Qf = Lq @ np.diag([10.0, 1.0, 0.5]) @ Lq.T, np.diag([0.2]), np.diag([20.0, 2.0, 1.0])
    return dict(T=T, Q=Q, R=R, Qf=Qf, alpha=1.5, tol=0.5, mu_u=mu_u, sig_u=0.5 * np.eye(1), x0=x0)


def run_both(lib, device, n_iters=4, inference="cubature", **kw):
    from oracle.i2c_numpy import CubatureRule, I2cOracle

    p = problem()
    Q, R = p["Q"], p["R"]
    model = make_env_model(SpringChainKnown())
    eng = pkg.BatchedI2c(model, p["T"], Q, R, p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"],
                         device=device, lib=lib, inference=inference, **kw)
    assert eng.model_id >= pkg._native.PLUGIN_BASE and (eng.nx, eng.nu, eng.nz, eng.nzt) == (3, 1, 4, 3)
    if inference == "linearize":
        from oracle.i2c_linearize_numpy import I2cLinearizeOracle

        ora = I2cLinearizeOracle(SpringChainNumpy(), p["T"], Q, R, p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"])
    else:
        ora = I2cOracle(SpringChainNumpy(), p["T"], Q, R, p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], rule=CubatureRule(1, 0, 0), x0=p["x0"])
    for it in range(1, n_iters + 1):
        eng.learn_msgs()
        ora.learn_msgs()
        what = f"spring chain ({inference}, {eng.forward_family}/{eng.backward_family}) it{it}"
        mu, sig = eng.marginal_state_action()
        close(np_(mu), ora.mu_xu0_m, 1e-8, what + " mu_xu0_m")
        close(np_(sig), ora.sig_xu0_m, 1e-8, what + " sig_xu0_m")
        f = eng.forward_messages()
        for k in ("mu_xu1_f", "sig_xu1_f", "mu_x3_f", "sig_x3_f"):
            if k in f and hasattr(ora, k):
                close(np_(f[k]), getattr(ora, k), 1e-8, what + " " + k)
        Kg, k, sigK = eng.local_linear_policy()
        close(np_(Kg), ora.K, 1e-7, what + " K")
        close(np_(k), ora.k, 1e-7, what + " k")
        close(np_(sigK), ora.sigK, 1e-7, what + " sigK")
        close(np_(eng.alpha), ora.alpha, 1e-8, what + " alpha")
        close(np_(eng.costs_m[-1]), ora.costs_m[-1], 1e-8, what + " cost")
    assert eng.failures() == []
    return eng


def build_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(ROOT, "input-inference-for-control_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def hostsim_lib():
    import hostsim

    return hostsim.load()


@pytest.mark.parametrize("lanes", [0, 4, 64], ids=["lane", "group", "quad"])
def test_identity_plugin_em_vs_oracle_hostsim(hostsim_lib, lanes):
    eng = run_both(hostsim_lib, "cpu", group_lanes=lanes)
    assert eng.forward_family == {0: "lane", 4: "group", 64: "quad"}[lanes]


def test_identity_plugin_linearize_vs_oracle_hostsim(hostsim_lib):
    run_both(hostsim_lib, "cpu", n_iters=3, inference="linearize")


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 4, 64], ids=["lane", "group", "quad"])
def test_identity_plugin_em_vs_oracle_gpu(lanes):
    eng = run_both(pkg.load_library(), "cuda", group_lanes=lanes)
    assert eng.forward_family == {0: "lane", 4: "group", 64: "quad"}[lanes]


@pytest.mark.gpu
def test_identity_plugin_linearize_vs_oracle_gpu():
    run_both(pkg.load_library(), "cuda", n_iters=3, inference="linearize")
