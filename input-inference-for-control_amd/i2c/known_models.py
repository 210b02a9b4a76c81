"""Known-model plugins (`sys` objects) for the MI355X build.

Each class carries (a) the attribute protocol the reference's callers read (dim_x, dim_u,
dim_xu, dim_z, dim_z_term, x0 (nx,1), sig_x0, sig_eta, zg (nz,1), zg_term, xu_lim ...;
reference i2c/env_def.py:34-82) and (b) ``model_id`` / ``device_params()`` selecting the
compile-time device functor in csrc/i2c_models.hpp that the solver actually evaluates.

The NumPy methods (``dynamics``, ``observe``, ``observe_terminal``, ``forward`` ...) exist for
the host-side callers of the same protocol (simulators, the MPC state estimator, scripts that
poke ``sys.forward`` directly): they are NOT used by I2cGraph, whose sweeps run on the GPU.
"""
import os

import numpy as np


class KnownModel:
    """Base of all known models (reference: BaseDef env_def.py:12-137 + BaseModelKnown model.py:144-183)."""

    name = "Template"
    model_name = None
    model_id = None
    data_driven = False
    deterministic = False
    model = None
    dim_x = dim_u = dim_z = dim_z_term = None
    xag = None
    x0_dist = None
    xu_lim = None

    def __init__(self, model=None, model_def=None):
        assert model is None and model_def is None, "only known models exist in the MI355X build"

    # ---- dimensions -----------------------------------------------------------------
    @property
    def dim_xu(self):
        return self.dim_x + self.dim_u

    dim_s = dim_xat = dim_xu

    @property
    def dim_yt(self):
        return self.dim_x

    @property
    def random_starting_state(self):
        return self.x0_dist is not None

    # ---- targets ----------------------------------------------------------------------
    @property
    def zg(self):
        zeros_u = np.zeros((self.dim_u, 1))
        return zeros_u if self.xag is None else np.vstack((self.xag, zeros_u))

    @property
    def zg_term(self):
        return self.zg

    # ---- device side ------------------------------------------------------------------
    # An OUT-OF-TREE model names the header with its device functor instead of a compiled-in
    # model_id: hip_header = path of the header, hip_struct = the functor's name in namespace i2c (default: the file name in
    # CamelCase), hip_name = the library's name (default: the file name). resolve_model_id() builds lib/libi2c_model_<name>.so
    # when it is missing or older than its sources, loads it into the solver library and returns the id it was given.
    hip_header = None
    hip_struct = None
    hip_name = None

    def device_params(self):
        return []

    def resolve_model_id(self, lib):
        """The I2cProblem.model_id of this model in `lib` (a _native.NativeLibrary): compiled in, or registered on first use."""
        if self.model_id is not None:
            return int(self.model_id)
        if self.hip_header is None:
            raise TypeError(f"{type(self).__name__}: a model needs a compiled-in model_id or a hip_header with its device functor "
                            "(arbitrary Python callables cannot run inside the kernels; INTEGRATION.md section 3)")
        ids = lib.__dict__.setdefault("_plugin_ids", {})
        key = (os.path.abspath(self.hip_header), self.hip_struct, self.hip_name)
        if key not in ids:
            ids[key] = lib.load_model(self._build_plugin(lib))[0]
        return ids[key]

    def _build_plugin(self, lib):
        import importlib.util

        pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(pkg_dir, "build.py"))
        build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(build)
        # (the host simulation of tests/ gets the g++ build of the same header, next to its own library)
        return build.build_model(self.hip_header, struct=self.hip_struct, name=self.hip_name, host_sim=lib.is_host_sim,
                                 out_dir=os.path.dirname(lib.path) if lib.is_host_sim else None, verbose=False)

    # ---- host-side protocol -----------------------------------------------------------
    def dynamics(self, xu):
        raise NotImplementedError

    def observe(self, xu):
        raise NotImplementedError

    def observe_terminal(self, x):
        raise NotImplementedError

    def observe_terminal_x(self, x):
        return self.observe_terminal(x)

    sig_zeta = None

    def measure(self, x):
        """Measurement of the MPC state estimator. The reference defines it only for its quadrotor; the
        other models observe their terminal features here (matches the device functor)."""
        return self.observe_terminal(x)

    def forward(self, xu):
        xn = self.dynamics(xu)
        return xn, np.broadcast_to(self.sig_eta, (xu.shape[0],) + self.sig_eta.shape).copy()

    def __call__(self, xu):
        return self.forward(xu)

    def predict(self, xu):
        return self.dynamics(xu)

    def init(self):
        return self.x0.squeeze(), self.sig_x0

    def sample(self, xu):
        mean, cov = self.forward(xu)
        noise = np.random.randn(xu.shape[0], self.dim_x)
        return mean + np.einsum("bij,bj->bi", np.linalg.cholesky(cov), noise)

    def clip_u(self, u):
        return np.clip(u, self.xu_lim[0, self.dim_x:], self.xu_lim[1, self.dim_x:])

    def run(self, horizon, policy, deterministic=False):
        """Roll a policy through the model (reference BaseModel.run, model.py:65-77)."""
        XU = np.zeros((horizon, self.dim_xu))
        Z = np.zeros((horizon, self.dim_z))
        x = np.array(self.x0, dtype=float).reshape(1, self.dim_x)
        for t in range(horizon):
            u = policy(t, x.T, deterministic=deterministic).T
            xu = np.hstack((x, u))
            XU[t], Z[t] = xu, self.observe(xu)[0]
            x = self.sample(xu)
        return XU, Z, np.squeeze(self.observe_terminal(x))

    def train(self, *a, **k):
        pass

    def save(self, path):
        print("Known model, no saving")

    # ---- host-side linearisations (reference: env_def.py observe_linearize / observe_terminal_linearize / dydxu of each model,
    # BaseModelKnown.forward_linearize model.py:158-164). The solver's Linearize() cells differentiate the DEVICE functors with dual
    # numbers; these are for callers that poke the model directly. Jacobians by central differences of the NumPy functions
    # (relative step 1e-6: ~1e-9 accurate), so every model -- out-of-tree ones included -- has them without writing derivatives.
    @staticmethod
    def _jacobian(f, x):
        x = np.asarray(x, float).reshape(1, -1)
        y0 = np.reshape(f(x), -1)
        jac = np.zeros((y0.size, x.shape[1]))
        for i in range(x.shape[1]):
            h = 1e-6 * max(1.0, abs(x[0, i]))
            e = np.zeros_like(x)
            e[0, i] = h
            jac[:, i] = (np.reshape(f(x + e), -1) - np.reshape(f(x - e), -1)) / (2.0 * h)
        return jac

    def dydxu(self, xu):
        """d dynamics / d [x; u] at one point, (dim_x, dim_xu)."""
        return self._jacobian(self.dynamics, xu)

    def forward_linearize(self, xu):
        """(x', A, B, a, sig_eta) with x' = A x + B u + a to first order about xu (1, dim_xu)."""
        xu = np.asarray(xu, float)
        assert xu.shape == (1, self.dim_xu)
        xn = np.reshape(self.dynamics(xu), (-1, 1))
        ab = self.dydxu(xu)
        return xn, ab[:, :self.dim_x], ab[:, self.dim_x:], xn - ab @ xu.T, self.sig_eta

    def observe_linearize(self, xu):
        """(z, C, c, D) with z = C x + D u + c to first order about xu (1, dim_xu)."""
        xu = np.asarray(xu, float).reshape(1, self.dim_xu)
        z = np.reshape(self.observe(xu), (-1, 1))
        cd = self._jacobian(self.observe, xu)
        return z, cd[:, :self.dim_x], z - cd @ xu.T, cd[:, self.dim_x:]

    def observe_terminal_linearize(self, x):
        """(z, C, c) with z = C x + c to first order about x (dim_x, 1) or (1, dim_x)."""
        x = np.asarray(x, float).reshape(1, self.dim_x)
        z = np.reshape(self.observe_terminal(x), (-1, 1))
        cj = self._jacobian(self.observe_terminal, x)
        return z, cj, z - cj @ x.T

    def predict_1d(self, x, u):
        return self.dynamics(np.hstack((np.reshape(x, -1), np.reshape(u, -1))).reshape(1, self.dim_xu))

    # ---- state / action limits (reference BaseDef, env_def.py:96-137) ------------------------------------------------------
    def remove_state_bounds(self):
        pass

    def xu_in_bounds(self, xu):
        lim = np.asarray(self.xu_lim, float)
        assert lim.shape == (2, self.dim_xu)
        return bool(np.all(lim[0, :, None] < xu) and np.all(lim[1, :, None] > xu))

    def x_in_bounds(self, xu):
        lim = np.asarray(self.xu_lim, float)
        assert lim.shape == (2, self.dim_xu)
        x = xu[:, :self.dim_x]
        return bool(np.all(lim[0, :self.dim_x, None] < x) and np.all(lim[1, :self.dim_x, None] > x))

    def filter_state_constraint_violations(self, xu, dx):
        """Cut an episode (rows of xu and dx) at the first state outside 99.9 % of its limits."""
        lim = np.asarray(self.xu_lim, float)
        x = xu[:, :self.dim_x]
        inside = np.clip(x, 0.999 * lim[0, :self.dim_x], 0.999 * lim[1, :self.dim_x])
        bad = np.flatnonzero(np.any(inside != x, axis=1))
        if bad.size == 0:
            return xu, dx
        return xu[:bad[0]], dx[:bad[0]]


_INF = np.inf


class PendulumKnown(KnownModel):
    """Torque-limited pendulum swing-up (reference env_def.py:233-309, env_autograd.py:5-19)."""

    name = "Pendulum"
    model_name = "PendulumKnown"
    model_id = 0
    key = ["$\\theta$", "$\\dot{\\theta}$", "$u$"]
    z_key = ["$\\sin(\\theta)$", "$\\cos(\\theta)$", "$\\dot{\\theta}$", "$u$"]
    unit = ["rad", "rad/s", "Nm"]
    dim_x, dim_u, dim_z, dim_z_term, dim_xuf = 2, 1, 4, 3, 4

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.x0 = np.array([[np.pi], [0.0]])
        self.xg = np.zeros((2, 1))
        self.xag = np.array([[0.0], [1.0], [0.0]])
        self._zg_term = np.array([[0.0], [1.0], [0.0]])
        self.sig_x0 = 1e-5 * np.eye(2)
        self.sig_eta = np.diag([1e-5, 1e-5])
        self.xu_lim = np.array([[-_INF, -_INF, -2.0], [_INF, _INF, 2.0]])

    @property
    def zg_term(self):
        return self._zg_term

    def dynamics(self, xu):
        xu = np.asarray(xu, dtype=float)
        th, om = xu[:, 0], xu[:, 1]
        torque = np.clip(xu[:, 2], -2.0, 2.0)
        acc = (-3.0 * 9.80665 / 2.0) * np.sin(th + np.pi) - 1e-2 * om + 3.0 * torque
        om2 = om + 0.05 * acc
        return np.column_stack((th + 0.05 * om2, om2))

    def observe(self, xu):
        return np.column_stack((np.sin(xu[:, 0]), np.cos(xu[:, 0]), xu[:, 1], xu[:, 2]))

    def observe_terminal(self, x):
        return np.column_stack((np.sin(x[:, 0]), np.cos(x[:, 0]), x[:, 1]))


class PendulumKnownActReg(PendulumKnown):
    """Pendulum with only the action in the cost (covariance control; env_def.py:312-346)."""

    model_name = "PendulumKnownActReg"
    model_id = 1
    z_key = ["$u$"]
    dim_z, dim_z_term = 1, 1

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.xag = None
        self.xg = np.array([[None], [None]])

    def observe(self, xu):
        return xu[:, self.dim_x:]

    def observe_terminal(self, x):
        return None

    def measure(self, x):
        return PendulumKnown.observe_terminal(self, x)


class CartpoleKnown(KnownModel):
    """Cart-pole swing-up (env_def.py:491-612, env_autograd.py:25-54)."""

    name = "Cartpole"
    model_name = "CartpoleKnown"
    model_id = 2
    dim_x, dim_u, dim_z, dim_z_term, dim_xuf = 4, 1, 6, 5, 6

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.x0 = np.array([[0.0], [np.pi], [0.0], [0.0]])
        self.xg = np.zeros((4, 1))
        self.xag = np.array([[0.0], [0.0], [1.0], [0.0], [0.0]])
        self._zg_term = self.xag.copy()
        self.sig_x0 = 1e-5 * np.eye(4)
        self.sig_eta = 1e-8 * np.eye(4)
        self.xu_lim = np.array([[-_INF] * 4 + [-5.0], [_INF] * 4 + [5.0]])

    @property
    def zg_term(self):
        return self._zg_term

    def dynamics(self, xu):
        xu = np.asarray(xu, dtype=float)
        g, m_p, ln, dt = 9.81, 0.127, 0.3365, 1.0 / 250.0
        m_t = 0.37 + m_p
        u = np.clip(xu[:, 4], -5.0, 5.0)
        s, c, w2 = np.sin(xu[:, 1]), np.cos(xu[:, 1]), xu[:, 3] ** 2
        th_acc = (-m_p * ln * s * c * w2 + m_t * g * s - u * c) / (ln * (4.0 / 3.0 * m_t - m_p * c ** 2))
        x_acc = (m_p * ln * s * w2 - m_p * ln * th_acc * c + u) / m_t
        return np.column_stack((xu[:, 0] + dt * xu[:, 2], xu[:, 1] + dt * xu[:, 3], xu[:, 2] + dt * x_acc,
                                xu[:, 3] + dt * th_acc))

    def observe(self, xu):
        return np.column_stack((xu[:, 0], np.sin(xu[:, 1]), np.cos(xu[:, 1]), xu[:, 2], xu[:, 3], xu[:, 4]))

    def observe_terminal(self, x):
        return np.column_stack((x[:, 0], np.sin(x[:, 1]), np.cos(x[:, 1]), x[:, 2], x[:, 3]))


class DoubleCartpoleKnown(KnownModel):
    """Double cart-pole swing-up (env_def.py:615-761, env_autograd.py:60-167)."""

    name = "Double Cartpole"
    model_name = "DoubleCartpoleKnown"
    model_id = 3
    dim_x, dim_u, dim_z, dim_z_term, dim_xuf = 6, 1, 9, 8, 9

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.x0 = np.array([[0.0], [np.pi], [np.pi], [0.0], [0.0], [0.0]])
        self.xg = np.zeros((6, 1))
        self.xag = np.array([[0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0]]).T
        self._zg_term = self.xag.copy()
        self.sig_x0 = 1e-6 * np.eye(6)
        self.sig_eta = 1e-6 * np.eye(6)
        self.xu_lim = np.array([[-_INF] * 6 + [-10.0], [_INF] * 6 + [10.0]])

    @property
    def zg_term(self):
        return self._zg_term

    def dynamics(self, xu):
        xu = np.asarray(xu, dtype=float)
        dt, g, m1, m2, L1 = 1 / 125, 9.81, 0.127, 0.127, 0.3365
        l1 = l2 = L1 / 2
        m_t = 0.37 + m1 + m2
        h1, h2, h3 = m1 * l1 + m2 * L1, m2 * l2, L1 * l2 * m2
        s1, c1, s2, c2 = np.sin(xu[:, 1]), np.cos(xu[:, 1]), np.sin(xu[:, 2]), np.cos(xu[:, 2])
        sd, cd = np.sin(xu[:, 1] - xu[:, 2]), np.cos(xu[:, 1] - xu[:, 2])
        n = xu.shape[0]
        M = np.empty((n, 3, 3))
        M[:, 0, 0] = m_t
        M[:, 0, 1] = M[:, 1, 0] = h1 * c1
        M[:, 0, 2] = M[:, 2, 0] = h2 * c2
        M[:, 1, 1] = l1 ** 2 * m1 + L1 ** 2 * m2 + m1 * L1 / 12
        M[:, 1, 2] = M[:, 2, 1] = h3 * cd
        M[:, 2, 2] = l2 ** 2 * m2 + m2 * L1 / 12
        w1, w2 = xu[:, 4], xu[:, 5]
        rhs = np.column_stack((
            3.0 * np.clip(xu[:, 6], -10.0, 10.0) + h1 * w1 * w1 * s1 + h2 * w2 * w2 * s2,
            -h3 * w2 * w2 * sd + h1 * g * s1,
            h3 * w1 * w1 * sd + h2 * g * s2,
        ))
        acc = np.linalg.solve(M, rhs[:, :, None])[:, :, 0]
        vel = xu[:, 3:6] + dt * acc
        return np.hstack((xu[:, 0:3] + dt * vel, vel))

    def observe(self, xu):
        return np.column_stack((xu[:, 0], np.sin(xu[:, 1]), np.cos(xu[:, 1]), np.sin(xu[:, 2]), np.cos(xu[:, 2]),
                                xu[:, 3], xu[:, 4], xu[:, 5], xu[:, 6]))

    def observe_terminal(self, x):
        return np.column_stack((x[:, 0], np.sin(x[:, 1]), np.cos(x[:, 1]), np.sin(x[:, 2]), np.cos(x[:, 2]),
                                x[:, 3], x[:, 4], x[:, 5]))


class LinearExact(KnownModel):
    """Linear dynamical system used for the LQR equivalence (env_def.py:139-191, model.py:226-246)."""

    name = "Linear"
    model_name = "LinearKnown"
    model_id = 4
    key = ["$x_1$", "$x_2$", "$u$"]
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 3, 2

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.x0 = np.array([[5.0], [5.0]])
        self.xg = np.array([[1.0], [-1.0]])
        self.xag = self.xg
        self.zg_term_ = self.xg
        self.sig_x0 = 1e-20 * np.eye(2)
        self.sig_eta = 1e-20 * np.eye(2)
        self.A = np.array([[1.1, 0.0], [0.1, 1.1]])
        self.B = np.array([[0.1], [0.0]])
        self.a = self.xg - self.A @ self.xg
        self.xu_lim = np.array([[-_INF] * 3, [_INF] * 3])

    @property
    def zg_term(self):
        return self.zg_term_

    @zg_term.setter
    def zg_term(self, v):
        self.zg_term_ = v

    @property
    def AB(self):
        return np.concatenate((self.A, self.B), axis=1)

    def device_params(self):
        return list(self.A.reshape(-1)) + list(self.B.reshape(-1)) + list(np.asarray(self.a).reshape(-1))

    def dynamics(self, xu):
        return xu @ self.AB.T + np.asarray(self.a).reshape(1, -1)

    def observe(self, xu):
        return np.array(xu, dtype=float)

    def observe_terminal(self, x):
        return np.array(x, dtype=float)


class LinearMinimumEnergy(LinearExact):
    """LDS with action-only cost for covariance control (env_def.py:194-230)."""

    model_name = "LinearKnownMinimumEnergy"
    model_id = 5
    dim_z = 1

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.sig_x0 = np.diag([1e-1, 5e0])
        self.xag = None
        self.zg_term_ = np.array([[-5.0], [-5.0]])
        self.A = np.array([[1.05, 0.0], [0.05, 1.01]])
        self.B = np.array([[0.1], [0.0]])
        self.a = self.zg_term_ - self.A @ self.zg_term_
        self.sig_eta = np.diag([1e-1, 1e-2])

    def observe(self, xu):
        return xu[:, self.dim_x:]


class PlanarQuadrotor(KnownModel):
    """Build-defined analytic planar quadrotor with the interface, dimensions and constants of the
    reference's Box2D QuadrotorDef (scripts/mpc_state_est/mpc_quad.py:219-383); see
    csrc/i2c_models.hpp for what is and is not reproducible."""

    name = "2D Quadrotor"
    model_name = "PlanarQuadrotor"
    model_id = 6
    dim_x, dim_u, dim_z, dim_y = 6, 2, 8, 8
    dim_z_term = 6
    W, H = 20.0, 40.0 / 3.0
    dt, arm, half_h, density, ang_damp, grav, force_mx = 0.1, 0.8, (40.0 / 3.0) / 100.0, 5.0, 0.5, 9.81, 30.0

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        w, h = 2 * self.arm, 2 * self.half_h
        self.mass = self.density * w * h
        self.inertia = self.mass * (w * w + h * h) / 12.0
        self.x0 = np.array([[self.W / 4, self.H / 2, 0.0, 0.0, 0.0, 0.0]]).T
        self.sig_x0 = 1e-5 * np.eye(6)
        self.sig_eta = np.diag([1e-6] * 3 + [1e-4] * 3)
        self.xag = np.array([[3 * self.W / 4, self.H / 2, 0.0, 0.0, 0.0, 0.0]]).T
        self.sig_zeta = None  # set by the experiment (mpc_quad.py:552-556)
        self.xu_lim = np.array([[-_INF] * 6 + [0.0, 0.0], [_INF] * 6 + [self.force_mx] * 2])

    @property
    def gravity(self):
        return self.grav * self.mass

    @property
    def zg_term(self):
        return self.xag

    def device_params(self):
        return [self.mass, self.inertia, self.force_mx]

    def dynamics(self, xu):
        xu = np.asarray(xu, dtype=float)
        f1, f2 = np.clip(xu[:, 6], 0.0, self.force_mx), np.clip(xu[:, 7], 0.0, self.force_mx)
        thrust = f1 + f2
        vx = xu[:, 3] - self.dt * thrust * np.sin(xu[:, 2]) / self.mass
        vy = xu[:, 4] + self.dt * (thrust * np.cos(xu[:, 2]) / self.mass - self.grav)
        om = (xu[:, 5] + self.dt * self.arm * (f2 - f1) / self.inertia) / (1.0 + self.dt * self.ang_damp)
        return np.column_stack((xu[:, 0] + self.dt * vx, xu[:, 1] + self.dt * vy, xu[:, 2] + self.dt * om, vx, vy, om))

    def observe(self, xu):
        return np.array(xu, dtype=float)

    def observe_terminal(self, x):
        return np.array(x, dtype=float)

    def measure(self, x):
        """Rotor-tip positions and velocities, with the reference's rxd / ryd expressions verbatim in meaning."""
        d, s, c, w = self.arm, np.sin(x[:, 2]), np.cos(x[:, 2]), x[:, 5]
        return np.column_stack((x[:, 0] - d * c, x[:, 1] - d * s, x[:, 0] + d * c, x[:, 1] + d * s,
                                x[:, 3] + d * s * w, x[:, 4] - d * c * w, x[:, 3] + d - s * w, x[:, 4] + d + c * w))


class Quadrotor12(KnownModel):
    """Build-defined 12-state, 4-rotor quadrotor (BASELINE config 4: nx = 12) with the plugin interface of the reference's
    QuadrotorDef (scripts/mpc_state_est/mpc_quad.py:219-383); see csrc/i2c_models.hpp. d = 16: no one-lane kernels; the sweeps run on the wave kernels (one wavefront per
    trajectory, csrc/i2c_wave.hpp), propagation and the filter on the group kernels."""

    name = "3D Quadrotor"
    model_name = "Quadrotor12"
    model_id = 7
    dim_x, dim_u, dim_z, dim_y = 12, 4, 16, 9
    dim_z_term = 12
    dt, arm, kq, ang_damp, grav = 0.05, 0.25, 0.05, 0.5, 9.81
    mass, Ixx, Iyy, Izz, force_mx = 1.0, 0.02, 0.02, 0.04, 6.0

    def __init__(self, model=None, model_def=None):
        super().__init__(model, model_def)
        self.x0 = np.zeros((12, 1))
        self.sig_x0 = 1e-5 * np.eye(12)
        self.sig_eta = np.diag([1e-6] * 6 + [1e-4] * 6)
        self.xag = np.array([[1.0, 1.0, 1.0] + [0.0] * 9]).T
        self.sig_zeta = None  # set by the experiment, as mpc_quad.py:552-556
        self.xu_lim = np.array([[-_INF] * 12 + [0.0] * 4, [_INF] * 12 + [self.force_mx] * 4])

    @property
    def gravity(self):
        return self.grav * self.mass

    @property
    def zg_term(self):
        return self.xag

    def device_params(self):
        return [self.mass, self.Ixx, self.Iyy, self.Izz, self.force_mx]

    def dynamics(self, xu):
        xu = np.asarray(xu, dtype=float)
        f1, f2, f3, f4 = (np.clip(xu[:, 12 + i], 0.0, self.force_mx) for i in range(4))
        thrust = (f1 + f2) + (f3 + f4)
        torque = (self.arm * (f2 - f4), self.arm * (f3 - f1), self.kq * ((f1 - f2) + (f3 - f4)))
        s, c = np.sin(xu[:, 3:6]), np.cos(xu[:, 3:6])
        w = xu[:, 9:12]
        damp = 1.0 / (1.0 + self.dt * self.ang_damp)
        inertia = (self.Ixx, self.Iyy, self.Izz)
        gyro = ((self.Izz - self.Iyy) * w[:, 1] * w[:, 2], (self.Ixx - self.Izz) * w[:, 2] * w[:, 0],
                (self.Iyy - self.Ixx) * w[:, 0] * w[:, 1])
        wn = [(w[:, i] + self.dt * (torque[i] - gyro[i]) / inertia[i]) * damp for i in range(3)]
        am = thrust / self.mass
        vn = [xu[:, 6] + self.dt * am * (c[:, 0] * s[:, 1] * c[:, 2] + s[:, 0] * s[:, 2]),
              xu[:, 7] + self.dt * am * (c[:, 0] * s[:, 1] * s[:, 2] - s[:, 0] * c[:, 2]),
              xu[:, 8] + self.dt * (am * (c[:, 0] * c[:, 1]) - self.grav)]
        mix = s[:, 0] * wn[1] + c[:, 0] * wn[2]
        rates = [wn[0] + (s[:, 1] / c[:, 1]) * mix, c[:, 0] * wn[1] - s[:, 0] * wn[2], mix / c[:, 1]]
        return np.column_stack([xu[:, i] + self.dt * vn[i] for i in range(3)]
                               + [xu[:, 3 + i] + self.dt * rates[i] for i in range(3)] + vn + wn)

    def observe(self, xu):
        return np.array(xu, dtype=float)

    def observe_terminal(self, x):
        return np.array(x, dtype=float)

    def measure(self, x):
        x = np.asarray(x, dtype=float)
        return np.column_stack((x[:, :6], x[:, 9:12]))


ENVIRONMENTS = {
    "LinearKnown": LinearExact,
    "LinearKnownMinimumEnergy": LinearMinimumEnergy,
    "PendulumKnown": PendulumKnown,
    "PendulumKnownActReg": PendulumKnownActReg,
    "CartpoleKnown": CartpoleKnown,
    "DoubleCartpoleKnown": DoubleCartpoleKnown,
    "PlanarQuadrotor": PlanarQuadrotor,
    "Quadrotor12": Quadrotor12,
}


def make_env_model(env_def, model_def=None):
    """Same call as the reference's i2c.model.make_env_model (model.py:19-44); known models only."""
    if model_def is not None:
        raise ValueError("learned models are not part of the MI355X build (none exists in the reference either)")
    if isinstance(env_def, KnownModel):  # an out-of-tree model object (hip_header): taken as it is
        return env_def
    if isinstance(env_def, type) and issubclass(env_def, KnownModel):
        return env_def()
    try:
        return ENVIRONMENTS[env_def]()
    except KeyError:
        raise KeyError(f"unknown or unsupported environment '{env_def}'; available: {sorted(ENVIRONMENTS)}")
