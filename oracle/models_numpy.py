"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy, fp64) restatement of the reference's known-model
plugins (`sys` objects). Never imported by the product path; only tests/, smoke() and
bench.py's cpu_baseline leg may use it.

Each model exposes the *downward-facing sys protocol* of SURVEY.md section 8(b), but
vectorised over arbitrary leading axes (the reference only has the sigma-point axis):

    dim_x, dim_u, dim_xu, dim_z, dim_z_term, x0, sig_x0, sig_eta, zg, zg_term
    dynamics(xu[..., d])          -> x'[..., nx]
    observe(xu[..., d])           -> z[..., nz]
    observe_terminal(x[..., nx])  -> zT[..., nzt]   (or None if the model has none)

Reference citations are relative to /root/reference.
Pinned against the real reference by tests/golden/models_*.npz (oracle/gen_golden.py).
"""
import numpy as np


class _Model:
    name = "base"
    model_id = -1
    dim_x = dim_u = dim_z = dim_z_term = 0
    has_terminal_obs = True

    @property
    def dim_xu(self):
        return self.dim_x + self.dim_u

    def observe_terminal(self, x):
        raise NotImplementedError


class Pendulum(_Model):
    """PendulumKnown: i2c/env_def.py:233-309 (definition), i2c/env_autograd.py:5-19 (step)."""

    name = "PendulumKnown"
    model_id = 0
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 4, 3

    def __init__(self):
        self.x0 = np.array([np.pi, 0.0])  # env_def.py:248
        self.sig_x0 = 1e-5 * np.eye(2)  # env_def.py:253
        self.sig_eta = 1e-5 * np.eye(2)  # env_def.py:257
        self.zg = np.array([0.0, 1.0, 0.0, 0.0])  # xag stacked on zero action, env_def.py:66-70,250
        self.zg_term = np.array([0.0, 1.0, 0.0])  # env_def.py:251

    def dynamics(self, xu):
        # env_autograd.py:5-19 : torque-limited, damped pendulum, semi-implicit Euler
        dt, mass, length, damp, grav, u_max = 0.05, 1.0, 1.0, 1e-2, 9.80665, 2.0
        th, om = xu[..., 0], xu[..., 1]
        u = np.clip(xu[..., 2], -u_max, u_max)
        acc = -3.0 * grav / (2 * length) * np.sin(th + np.pi) - damp * om
        acc = acc + 3.0 / (mass * length ** 2) * u
        om_new = om + acc * dt
        th_new = th + om_new * dt
        return np.stack((th_new, om_new), axis=-1)

    def observe(self, xu):
        # env_def.py:273-276
        return np.stack((np.sin(xu[..., 0]), np.cos(xu[..., 0]), xu[..., 1], xu[..., 2]), axis=-1)

    def observe_terminal(self, x):
        # env_def.py:288-291
        return np.stack((np.sin(x[..., 0]), np.cos(x[..., 0]), x[..., 1]), axis=-1)


class PendulumActReg(Pendulum):
    """PendulumKnownActReg: i2c/env_def.py:312-346 -- only the action is 'observed'."""

    name = "PendulumKnownActReg"
    model_id = 1
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 1, 1
    has_terminal_obs = False

    def __init__(self):
        super().__init__()
        self.zg = np.zeros(1)  # xag is None -> zero action target, env_def.py:66-70,318
        self.zg_term = np.zeros(1)

    def observe(self, xu):
        return xu[..., 2:]  # env_def.py:330-332

    def observe_terminal(self, x):
        return None  # env_def.py:342-343


class Cartpole(_Model):
    """CartpoleKnown: i2c/env_def.py:491-612, i2c/env_autograd.py:25-54."""

    name = "CartpoleKnown"
    model_id = 2
    dim_x, dim_u, dim_z, dim_z_term = 4, 1, 6, 5

    def __init__(self):
        self.x0 = np.array([0.0, np.pi, 0.0, 0.0])  # env_def.py:590
        self.sig_x0 = 1e-5 * np.eye(4)  # env_def.py:515
        self.sig_eta = 1e-8 * np.eye(4)  # env_def.py:594 (overrides :518)
        self.zg = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 0.0])  # env_def.py:592 + zero action
        self.zg_term = np.array([0.0, 0.0, 1.0, 0.0, 0.0])  # env_def.py:593

    def dynamics(self, xu):
        # env_autograd.py:25-54
        grav, m_cart, m_pole, length, dt, u_max = 9.81, 0.37, 0.127, 0.3365, 1 / 250.0, 5.0
        m_tot = m_cart + m_pole
        pos, th, vel, om = xu[..., 0], xu[..., 1], xu[..., 2], xu[..., 3]
        u = np.clip(xu[..., 4], -u_max, u_max)
        om2 = np.power(om, 2)
        s, c = np.sin(th), np.cos(th)
        num = -m_pole * length * s * c * om2 + m_tot * grav * s - u * c
        den = length * ((4.0 / 3.0) * m_tot - m_pole * c ** 2)
        th_acc = num / den
        x_acc = (m_pole * length * s * om2 - m_pole * length * th_acc * c + u) / m_tot
        return np.stack((pos + dt * vel, th + dt * om, vel + dt * x_acc, om + dt * th_acc), axis=-1)

    def observe(self, xu):
        # env_def.py:537-549
        return np.stack(
            (xu[..., 0], np.sin(xu[..., 1]), np.cos(xu[..., 1]), xu[..., 2], xu[..., 3], xu[..., 4]),
            axis=-1,
        )

    def observe_terminal(self, x):
        # env_def.py:567-570
        return np.stack((x[..., 0], np.sin(x[..., 1]), np.cos(x[..., 1]), x[..., 2], x[..., 3]), axis=-1)


class DoubleCartpole(_Model):
    """DoubleCartpoleKnown: i2c/env_def.py:615-761, i2c/env_autograd.py:60-167."""

    name = "DoubleCartpoleKnown"
    model_id = 3
    dim_x, dim_u, dim_z, dim_z_term = 6, 1, 9, 8

    def __init__(self):
        self.x0 = np.array([0.0, np.pi, np.pi, 0.0, 0.0, 0.0])  # env_def.py:648
        self.sig_x0 = 1e-6 * np.eye(6)  # env_def.py:654
        self.sig_eta = 1e-6 * np.eye(6)  # env_def.py:656
        self.zg_term = np.array([0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0])  # env_def.py:650,653
        self.zg = np.concatenate((self.zg_term, np.zeros(1)))  # env_def.py:651

    def dynamics(self, xu):
        # env_autograd.py:60-167 : M(q) qdd = B u - C(q,qd) qd - G(q), then semi-implicit Euler.
        dt, grav = 1 / 125, 9.81
        m_c, m1, m2 = 0.37, 0.127, 0.127
        m_tot = m_c + m1 + m2
        L1 = L2 = 0.3365
        l1, l2 = L1 / 2, L2 / 2
        J1, J2 = m1 * L1 / 12, m2 * L2 / 12
        u_max, gear = 10.0, 3.0

        th1, th2 = xu[..., 1], xu[..., 2]
        qd = xu[..., 3:6]
        s1, c1, s2, c2 = np.sin(th1), np.cos(th1), np.sin(th2), np.cos(th2)
        sd, cd = np.sin(th1 - th2), np.cos(th1 - th2)

        h1 = m1 * l1 + m2 * L2
        h2 = m2 * l2
        h3 = L1 * l2 * m2
        one = np.ones_like(th1)
        zero = np.zeros_like(th1)
        M = np.stack(
            (
                np.stack((m_tot * one, h1 * c1, h2 * c2), axis=-1),
                np.stack((h1 * c1, (l1 ** 2 * m1 + L1 ** 2 * m2 + J1) * one, h3 * cd), axis=-1),
                np.stack((h2 * c2, h3 * cd, (l2 ** 2 * m2 + J2) * one), axis=-1),
            ),
            axis=-2,
        )
        C = np.stack(
            (
                np.stack((zero, -h1 * qd[..., 1] * s1, -h2 * qd[..., 2] * s2), axis=-1),
                np.stack((zero, zero, h3 * qd[..., 2] * sd), axis=-1),
                np.stack((zero, -h3 * qd[..., 1] * sd, zero), axis=-1),
            ),
            axis=-2,
        )
        G = np.stack((zero, -(m1 * l1 + m2 * L1) * grav * s1, -m2 * l2 * grav * s2), axis=-1)
        u = gear * np.clip(xu[..., 6], -u_max, u_max)
        force = np.stack((u, zero, zero), axis=-1)
        rhs = force - np.einsum("...ij,...j->...i", C, qd) - G
        # the reference forms inv(M) and multiplies (env_autograd.py:157-160); keep that order
        qdd = np.einsum("...ij,...j->...i", np.linalg.inv(M), rhs)
        qd_new = qd + qdd * dt
        q_new = xu[..., 0:3] + qd_new * dt
        return np.concatenate((q_new, qd_new), axis=-1)

    def observe(self, xu):
        # env_def.py:682-695
        return np.stack(
            (
                xu[..., 0],
                np.sin(xu[..., 1]),
                np.cos(xu[..., 1]),
                np.sin(xu[..., 2]),
                np.cos(xu[..., 2]),
                xu[..., 3],
                xu[..., 4],
                xu[..., 5],
                xu[..., 6],
            ),
            axis=-1,
        )

    def observe_terminal(self, x):
        # env_def.py:719-732
        return np.stack(
            (
                x[..., 0],
                np.sin(x[..., 1]),
                np.cos(x[..., 1]),
                np.sin(x[..., 2]),
                np.cos(x[..., 2]),
                x[..., 3],
                x[..., 4],
                x[..., 5],
            ),
            axis=-1,
        )


class Linear(_Model):
    """LinearKnown: i2c/env_def.py:139-191, i2c/model.py:226-246.

    The shipped sig_x0 = sig_eta = 1e-20 I makes the cubature rule cancel catastrophically
    (SURVEY.md section 3.3); the constructor therefore takes the noise level as an argument
    (default = shipped value) so that fixtures can use a non-degenerate 1e-4.
    """

    name = "LinearKnown"
    model_id = 4
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 3, 2

    def __init__(self, noise=1e-20, goal=None):
        self.x0 = np.array([5.0, 5.0])
        # scripts/lqr_compare.py:128-131 redefines the goal (xag, zg_term, a) on the instance: `goal`
        self.xg = np.array([1.0, -1.0]) if goal is None else float(goal) * np.ones(2)
        self.sig_x0 = noise * np.eye(2)
        self.sig_eta = noise * np.eye(2)
        self.A = np.array([[1.1, 0.0], [0.1, 1.1]])
        self.B = np.array([[0.1], [0.0]])
        self.a = self.xg - self.A @ self.xg
        self.zg = np.concatenate((self.xg, np.zeros(1)))
        self.zg_term = self.xg.copy()

    @property
    def AB(self):
        return np.concatenate((self.A, self.B), axis=1)

    def dynamics(self, xu):
        return xu @ self.AB.T + self.a  # model.py:230-231

    def observe(self, xu):
        return xu + 0.0  # env_def.py:175-177 (c = 0)

    def observe_terminal(self, x):
        return x + 0.0  # env_def.py:183-185


class LinearMinimumEnergy(Linear):
    """LinearKnownMinimumEnergy: i2c/env_def.py:194-230 (covariance-control LDS)."""

    name = "LinearKnownMinimumEnergy"
    model_id = 5
    dim_x, dim_u, dim_z, dim_z_term = 2, 1, 1, 2

    def __init__(self):
        self.x0 = np.array([5.0, 5.0])
        self.sig_x0 = np.diag([1e-1, 5e0])
        self.zg_term = np.array([-5.0, -5.0])
        self.A = np.array([[1.05, 0.0], [0.05, 1.01]])
        self.B = np.array([[0.1], [0.0]])
        self.a = self.zg_term - self.A @ self.zg_term
        self.sig_eta = np.diag([1e-1, 1e-2])
        self.zg = np.zeros(1)

    def observe(self, xu):
        return xu[..., 2:]  # env_def.py:219-220


class PlanarQuadrotor(_Model):
    """BUILD-DEFINED analytic planar quadrotor (dynamics parity with the reference is UNPINNED).

    The reference's quadrotor (scripts/mpc_state_est/mpc_quad.py:219-383) steps a Box2D rigid
    body; Box2D is a third-party C++ engine that is neither vendored, pinned nor installed
    (SURVEY.md section 8c), so its arithmetic cannot be reproduced. This model keeps the
    reference's *interface, dimensions and constants* (dim_x=6: x, y, th, xd, yd, thd; dim_u=2
    rotor thrusts in [0, 30]; dim_z=8; measure() -> 8; FS = 10 -> dt = 0.1; W = 20, H = 40/3;
    body 2 vehicle_dx x 2 vehicle_dy = 1.6 x 0.2667 m at density 5 -> m = 2.133 kg; arm =
    vehicle_dx = 0.8 m; angularDamping 0.5; g = 9.81; x0 = [W/4, H/2, 0...], goal [3W/4, H/2, 0...])
    with Box2D's integrator for a free body: velocities first (semi-implicit Euler, damping as
    1 / (1 + dt c)), then positions.
    Solver parity for config 4 is pinned by feeding THIS model to the real reference
    I2cGraph / PartiallyObservedMpcPolicy in-container (oracle/gen_golden.py).
    """

    name = "PlanarQuadrotor"
    model_id = 6
    dim_x, dim_u, dim_z, dim_z_term, dim_y = 6, 2, 8, 6, 8
    W, H = 20.0, 40.0 / 3.0
    dt = 0.1
    arm = 20.0 / 25.0
    half_h = (40.0 / 3.0) / 100.0
    density = 5.0
    ang_damp = 0.5
    grav = 9.81
    u_max = 30.0

    def __init__(self):
        w, h = 2 * self.arm, 2 * self.half_h
        self.mass = self.density * w * h
        self.inertia = self.mass * (w * w + h * h) / 12.0
        self.gravity = self.grav * self.mass  # mpc_quad.py:321-323
        self.x0 = np.array([self.W / 4, self.H / 2, 0.0, 0.0, 0.0, 0.0])  # mpc_quad.py:243
        self.sig_x0 = 1e-5 * np.eye(6)  # mpc_quad.py:244
        self.sig_eta = np.diag([1e-6] * 3 + [1e-4] * 3)  # mpc_quad.py:245
        self.zg_term = np.array([3 * self.W / 4, self.H / 2, 0.0, 0.0, 0.0, 0.0])  # mpc_quad.py:250-251
        self.zg = np.concatenate((self.zg_term, np.zeros(2)))
        self.sig_zeta = np.diag([1e-6] * 8)  # low-noise setting, mpc_quad.py:552-554

    def dynamics(self, xu):
        px, py, th, vx, vy, om = (xu[..., i] for i in range(6))
        f1 = np.clip(xu[..., 6], 0.0, self.u_max)
        f2 = np.clip(xu[..., 7], 0.0, self.u_max)
        thrust = f1 + f2
        ax = -thrust * np.sin(th) / self.mass
        ay = thrust * np.cos(th) / self.mass - self.grav
        alpha = self.arm * (f2 - f1) / self.inertia
        vx_n = vx + self.dt * ax
        vy_n = vy + self.dt * ay
        om_n = (om + self.dt * alpha) / (1.0 + self.dt * self.ang_damp)
        return np.stack(
            (px + self.dt * vx_n, py + self.dt * vy_n, th + self.dt * om_n, vx_n, vy_n, om_n), axis=-1
        )

    def observe(self, xu):
        return xu + 0.0

    def observe_terminal(self, x):
        return x + 0.0

    def measure(self, x):
        """mpc_quad.py:370-383, including the reference's operator slip in rxd / ryd."""
        dx = self.arm
        s, c = np.sin(x[..., 2]), np.cos(x[..., 2])
        return np.stack(
            (
                x[..., 0] - dx * c,
                x[..., 1] - dx * s,
                x[..., 0] + dx * c,
                x[..., 1] + dx * s,
                x[..., 3] - dx * -s * x[..., 5],
                x[..., 4] - dx * c * x[..., 5],
                x[..., 3] + dx - s * x[..., 5],
                x[..., 4] + dx + c * x[..., 5],
            ),
            axis=-1,
        )


class Quadrotor12(_Model):
    """BUILD-DEFINED 12-state, 4-rotor quadrotor (BASELINE config 4 names nx = 12; the reference only has the planar
    Box2D body of scripts/mpc_state_est/mpc_quad.py:219-383, so dynamics parity is UNPINNED by construction).

    x = [p (3) | roll, pitch, yaw | v (3, world frame) | body rates (3)], u = four rotor thrusts clipped to [0, u_max],
    "+" configuration: roll torque arm (f2 - f4), pitch torque arm (f3 - f1), yaw torque kq (f1 - f2 + f3 - f4).
    Integrator in the order of the planar model (Box2D's for a free body): velocities first (semi-implicit Euler, angular
    damping as 1 / (1 + dt c)), then positions and Euler angles with the NEW velocities. observe / observe_terminal are the
    identity (as mpc_quad.py:355-368), measure = [p | angles | body rates].
    Solver parity is pinned by feeding THIS model to the real reference I2cGraph / PartiallyObservedMpcPolicy
    in-container (oracle/gen_golden.py: em_quad12_T20, mpc_quad12_fb).
    """

    name = "Quadrotor12"
    model_id = 7
    dim_x, dim_u, dim_z, dim_z_term, dim_y = 12, 4, 16, 12, 9
    dt, arm, kq, ang_damp, grav = 0.05, 0.25, 0.05, 0.5, 9.81
    mass, Ixx, Iyy, Izz, u_max = 1.0, 0.02, 0.02, 0.04, 6.0

    def __init__(self):
        self.gravity = self.grav * self.mass
        self.x0 = np.zeros(12)
        self.sig_x0 = 1e-5 * np.eye(12)
        self.sig_eta = np.diag([1e-6] * 6 + [1e-4] * 6)
        self.zg_term = np.array([1.0, 1.0, 1.0] + [0.0] * 9)
        self.zg = np.concatenate((self.zg_term, np.zeros(4)))
        self.sig_zeta = 1e-6 * np.eye(9)

    def dynamics(self, xu):
        f = [np.clip(xu[..., 12 + i], 0.0, self.u_max) for i in range(4)]
        thrust = (f[0] + f[1]) + (f[2] + f[3])
        tx, ty, tz = self.arm * (f[1] - f[3]), self.arm * (f[2] - f[0]), self.kq * ((f[0] - f[1]) + (f[2] - f[3]))
        sph, cph = np.sin(xu[..., 3]), np.cos(xu[..., 3])
        sth, cth = np.sin(xu[..., 4]), np.cos(xu[..., 4])
        sps, cps = np.sin(xu[..., 5]), np.cos(xu[..., 5])
        wx, wy, wz = xu[..., 9], xu[..., 10], xu[..., 11]
        damp = 1.0 / (1.0 + self.dt * self.ang_damp)
        wxn = (wx + self.dt * (tx - (self.Izz - self.Iyy) * wy * wz) / self.Ixx) * damp
        wyn = (wy + self.dt * (ty - (self.Ixx - self.Izz) * wz * wx) / self.Iyy) * damp
        wzn = (wz + self.dt * (tz - (self.Iyy - self.Ixx) * wx * wy) / self.Izz) * damp
        am = thrust / self.mass
        vxn = xu[..., 6] + self.dt * am * (cph * sth * cps + sph * sps)
        vyn = xu[..., 7] + self.dt * am * (cph * sth * sps - sph * cps)
        vzn = xu[..., 8] + self.dt * (am * (cph * cth) - self.grav)
        tth = sth / cth
        roll_d = wxn + tth * (sph * wyn + cph * wzn)
        pitch_d = cph * wyn - sph * wzn
        yaw_d = (sph * wyn + cph * wzn) / cth
        return np.stack(
            (xu[..., 0] + self.dt * vxn, xu[..., 1] + self.dt * vyn, xu[..., 2] + self.dt * vzn,
             xu[..., 3] + self.dt * roll_d, xu[..., 4] + self.dt * pitch_d, xu[..., 5] + self.dt * yaw_d,
             vxn, vyn, vzn, wxn, wyn, wzn), axis=-1)

    def observe(self, xu):
        return xu + 0.0

    def observe_terminal(self, x):
        return x + 0.0

    def measure(self, x):
        return np.concatenate((x[..., :6], x[..., 9:12]), axis=-1)


def _terminal_features_as_measurement(cls):
    """The reference defines `measure` only for its quadrotor. For the other models the build's state
    estimator observes the terminal features (no action), e.g. [sin th, cos th, thd]."""

    def measure(self, x):
        return self.observe_terminal(x)

    cls.measure = measure
    return cls


for _c in (Pendulum, Cartpole, DoubleCartpole, Linear, LinearMinimumEnergy):
    _terminal_features_as_measurement(_c)
PendulumActReg.measure = lambda self, x: Pendulum.observe_terminal(self, x)

MODELS = {
    m.name: m
    for m in (Pendulum, PendulumActReg, Cartpole, DoubleCartpole, Linear, LinearMinimumEnergy, PlanarQuadrotor,
              Quadrotor12)
}


def make_model(name, **kw):
    return MODELS[name](**kw)
