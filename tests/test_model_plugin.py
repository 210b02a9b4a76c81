"""The open model boundary (round-4 review, missing #1): a model that is NOT one of the eight compiled into libi2c_hip.so is
built from one header (tests/plugins/van_der_pol.hpp) into a library of its own, registered at run time through the C ABI
(i2c_load_model / i2c_register_model, ABI v7) and solved by the same kernels -- compared with the CPU oracle fed the same model in
NumPy. The reference's counterpart: any object with dim_*, forward, observe, observe_terminal_x is a model
(i2c/model.py:19-44, 154-156; i2c/env_def.py:34-82, 233-298).
CPU: the host simulation of the kernels (g++ build of the same header); `-m gpu`: the hipcc build on cuda:0."""
import ctypes as C
import importlib
import importlib.util
import os

import numpy as np
import pytest
import torch

from golden_util import assert_close
from parity import close, np_

pkg = importlib.import_module("input-inference-for-control_amd")
from i2c.known_models import KnownModel, make_env_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "tests", "plugins", "van_der_pol.hpp")
MU, DT, U_MAX = 1.5, 0.05, 3.0


class VanDerPolNumpy:
    """NumPy twin of tests/plugins/van_der_pol.hpp in the oracle's protocol (oracle/models_numpy.py): vectorised over leading axes."""

    name = "VanDerPol"
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 4, 2
    has_terminal_obs = True

    def __init__(self):
        self.x0 = np.array([1.0, 0.0])
        self.sig_x0 = 1e-4 * np.eye(2)
        self.sig_eta = 1e-5 * np.eye(2)
        self.zg = np.zeros(4)
        self.zg_term = np.zeros(2)

    @property
    def dim_xu(self):
        return 3

    def dynamics(self, xu):
        u = np.clip(xu[..., 2], -U_MAX, U_MAX)
        v = xu[..., 1] + DT * (MU * (1.0 - xu[..., 0] ** 2) * xu[..., 1] - xu[..., 0] + u)
        return np.stack((xu[..., 0] + DT * v, v), axis=-1)

    def observe(self, xu):
        return np.stack((xu[..., 0], xu[..., 1], xu[..., 1] / (1.0 + xu[..., 1] ** 2), xu[..., 2]), axis=-1)

    def observe_terminal(self, x):
        return x[..., :2]


class VanDerPolKnown(KnownModel):
    """The product-side plugin object: the reference's attribute protocol + the header of the device functor (no model_id)."""

    name = "VanDerPol"
    model_name = "VanDerPolKnown"
    hip_header = HEADER  # struct VanDerPol, library van_der_pol: the defaults derived from the file name
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 4, 2

    def __init__(self):
        super().__init__()
        n = VanDerPolNumpy()
        self._np = n
        self.x0 = n.x0.reshape(2, 1)
        self.xag = np.zeros((3, 1))
        self.sig_x0, self.sig_eta = n.sig_x0, n.sig_eta
        self.xu_lim = np.array([[-np.inf, -np.inf, -U_MAX], [np.inf, np.inf, U_MAX]])

    @property
    def zg(self):
        return np.zeros((4, 1))

    @property
    def zg_term(self):
        return np.zeros((2, 1))

    def device_params(self):
        return [MU, DT, U_MAX]

    def dynamics(self, xu):
        return self._np.dynamics(np.asarray(xu, float))

    def observe(self, xu):
        return self._np.observe(np.asarray(xu, float))

    def observe_terminal(self, x):
        return self._np.observe_terminal(np.asarray(x, float))


def problem(B=5, T=40, seed=3):
    rng = np.random.default_rng(seed)
    x0 = np.array([1.0, 0.0]) + 5e-2 * rng.normal(size=(B, 2))
    mu_u = 1e-2 * rng.normal(size=(B, T, 1))
    Q, R, Qf = np.diag([10.0, 1.0, 5.0]), np.diag([0.5]), np.diag([20.0, 2.0])
    return dict(T=T, Q=Q, R=R, Qf=Qf, alpha=2.0, tol=0.5, mu_u=mu_u, sig_u=0.5 * np.eye(1), x0=x0)


def run_both(lib, device, n_iters=5, inference="cubature", **kw):
    from oracle.i2c_numpy import CubatureRule, I2cOracle

    p = problem()
    model = make_env_model(VanDerPolKnown())
    eng = pkg.BatchedI2c(model, p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"],
                         device=device, lib=lib, inference=inference, **kw)
    assert eng.model_id >= pkg._native.PLUGIN_BASE and (eng.nx, eng.nu, eng.nz, eng.nzt) == (2, 1, 4, 2)
    if inference == "linearize":
        from oracle.i2c_linearize_numpy import I2cLinearizeOracle

        ora = I2cLinearizeOracle(VanDerPolNumpy(), p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"])
    else:
        ora = I2cOracle(VanDerPolNumpy(), p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"],
                        rule=CubatureRule(1, 0, 0), x0=p["x0"])
    for it in range(1, n_iters + 1):
        eng.learn_msgs()
        ora.learn_msgs()
        what = f"van der pol ({inference}, {eng.forward_family}/{eng.backward_family}) it{it}"
        mu, sig = eng.marginal_state_action()
        close(np_(mu), ora.mu_xu0_m, 1e-8, what + " mu_xu0_m")
        close(np_(sig), ora.sig_xu0_m, 1e-8, what + " sig_xu0_m")
        K, k, sigK = eng.local_linear_policy()
        close(np_(K), ora.K, 1e-7, what + " K")
        close(np_(k), ora.k, 1e-7, what + " k")
        close(np_(sigK), ora.sigK, 1e-7, what + " sigK")
        close(np_(eng.alpha), ora.alpha, 1e-8, what + " alpha")
        close(np_(eng.costs_m[-1]), ora.costs_m[-1], 1e-8, what + " cost")
    assert eng.failures() == []
    # the controller did something: the smoothed oscillation ends nearer the origin than the open-loop one starts
    return eng


def check_abi(lib, model_lib):
    """The registration entry points themselves, through ctypes as a foreign caller would."""
    N = pkg._native
    d = N.I2cDims()
    mid = lib.i2c_load_model(os.fsencode(model_lib), C.byref(d))
    assert mid >= N.PLUGIN_BASE and (d.nx, d.nu, d.nz, d.nzt, d.n_params, d.ny) == (2, 1, 4, 2, 3, 2)
    assert d.group_lanes == 4 and d.quad == 1 and d.wave == 0 and d.group_only == 0
    assert lib.i2c_load_model(os.fsencode(model_lib), None) == mid  # the same library again: the same id
    # i2c_register_model alone, for a caller that opened the model library itself
    dll = C.CDLL(model_lib)
    dll.i2c_model_ops.restype, dll.i2c_model_ops.argtypes = C.c_void_p, [C.c_int]
    dll.i2c_model_name.restype = C.c_char_p
    assert dll.i2c_model_abi_version() == N.ABI_VERSION and dll.i2c_model_name() == b"van_der_pol"
    ops = [dll.i2c_model_ops(t) for t in (N.F64, N.F32, N.F64_F32S)]
    assert all(ops) and dll.i2c_model_ops(7) is None
    assert lib.i2c_register_model(N.ABI_VERSION, ops[0], ops[1], ops[2], None) == mid
    assert lib.i2c_register_model(N.ABI_VERSION - 1, ops[0], ops[1], ops[2], None) == -1  # I2C_EINVAL: ABI mismatch
    assert lib.i2c_register_model(N.ABI_VERSION, None, None, None, None) == -1
    assert lib.i2c_load_model(b"/nonexistent/libi2c_model_x.so", None) == -1
    assert lib.i2c_load_model(os.fsencode(lib.path), None) == -1  # a library without the model symbols
    q = lib.query(mid)
    assert (q.nx, q.e_post, q.e_fwd) == (2, d.e_post, d.e_fwd)
    with pytest.raises(ValueError):
        lib.query(mid + 17)  # an id nobody was given
    p = N.I2cProblem()
    p.abi_version, p.model_id, p.B, p.T, p.quad_alpha = N.ABI_VERSION, mid, 64, 40, 1.0
    assert lib.i2c_backward_schedule(C.byref(p)) == N.BWD_CHUNKED and lib.i2c_workspace_bytes(mid, N.F64, 64, 40) > 0
    assert lib.i2c_kernel_family(C.byref(p), N.SWEEP_FORWARD) == N.FAMILY_LANE


def build_module():
    spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(ROOT, "input-inference-for-control_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- CPU: host simulation ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hostsim_lib():
    import hostsim

    return hostsim.load()


def test_plugin_is_not_in_the_tree():
    """Nothing under the package or include/ names the plugin: it is built and loaded from its header alone."""
    import subprocess

    r = subprocess.run(["grep", "-rIl", "-i", "-e", "van_der_pol", "-e", "VanDerPol", os.path.join(ROOT, "input-inference-for-control_amd", "csrc"),
                        os.path.join(ROOT, "input-inference-for-control_amd", "i2c"), os.path.join(ROOT, "input-inference-for-control_amd", "engine.py"),
                        os.path.join(ROOT, "input-inference-for-control_amd", "_native.py"), os.path.join(ROOT, "include")], capture_output=True, text=True)
    assert r.stdout.strip() == "", r.stdout


def test_plugin_abi_hostsim(hostsim_lib):
    model_lib = build_module().build_model(HEADER, host_sim=True, out_dir=os.path.dirname(hostsim_lib.path), verbose=False)
    check_abi(hostsim_lib, model_lib)


@pytest.mark.parametrize("lanes", [0, 4, 64], ids=["lane", "group", "quad"])
def test_plugin_em_vs_oracle_hostsim(hostsim_lib, lanes):
    eng = run_both(hostsim_lib, "cpu", group_lanes=lanes)
    assert eng.forward_family == {0: "lane", 4: "group", 64: "quad"}[lanes]


def test_plugin_linearize_vs_oracle_hostsim(hostsim_lib):
    run_both(hostsim_lib, "cpu", n_iters=3, inference="linearize")


def test_plugin_through_the_graph_api_hostsim(hostsim_lib):
    """The reference's own call: I2cGraph(sys, ...) with the plugin object as `sys` (B = 1), learn_msgs, the getters."""
    from i2c.exp_types import CubatureQuadrature
    from i2c.i2c import I2cGraph

    p = problem(B=1)
    g = I2cGraph(VanDerPolKnown(), p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"][0], p["sig_u"], None, None,
                 CubatureQuadrature(1, 0, 0), lib=hostsim_lib, device="cpu")
    for _ in range(3):
        g.learn_msgs()
    x, u = g.get_marginal_trajectory()[:2] if hasattr(g, "get_marginal_trajectory") else (None, None)
    assert len(g.costs_m) == 3 and np.isfinite(g.costs_m[-1])
    assert x is None or np.all(np.isfinite(x))


def test_model_without_id_or_header_is_refused(hostsim_lib):
    class Nothing(KnownModel):
        dim_x, dim_u, dim_z, dim_z_term = 2, 1, 4, 2

    with pytest.raises(TypeError):
        Nothing().resolve_model_id(hostsim_lib)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_plugin_abi_gpu():
    lib = pkg.load_library()
    check_abi(lib, build_module().build_model(HEADER, verbose=False))


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 4, 64], ids=["lane", "group", "quad"])
def test_plugin_em_vs_oracle_gpu(lanes):
    eng = run_both(pkg.load_library(), "cuda", group_lanes=lanes)
    assert eng.forward_family == {0: "lane", 4: "group", 64: "quad"}[lanes]


@pytest.mark.gpu
def test_plugin_linearize_vs_oracle_gpu():
    run_both(pkg.load_library(), "cuda", n_iters=3, inference="linearize")


@pytest.mark.gpu
def test_plugin_mixed_precision_and_batch_gpu():
    """fp32-stored messages (I2C_F64_F32S) and a batch that fills the chip, on the plugin's own translation units."""
    p = problem(B=4096, T=40)
    model = VanDerPolKnown()
    e64 = pkg.BatchedI2c(model, p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"], device="cuda")
    e32 = pkg.BatchedI2c(model, p["T"], p["Q"], p["R"], p["Qf"], p["alpha"], p["tol"], p["mu_u"], p["sig_u"], x0=p["x0"], device="cuda",
                         storage_dtype=torch.float32)
    e64.learn(3)
    e32.learn(3)
    torch.cuda.synchronize()
    assert e64.failures() == [] and e32.failures() == []
    assert_close(np_(e32.marginal_state_action()[0]), np_(e64.marginal_state_action()[0]), 1e-4, "fp32-stored vs fp64 means")
