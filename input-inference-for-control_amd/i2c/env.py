"""Simulators with the reference's interface (i2c/env.py): `make_env(experiment)` returns an object with
`simulated`, `run`, `batch_eval`, `plot_sim`, `run_render`, `close`, on top of the known-model definition.

`batch_eval(policy, n_eval)` is the reference's per-iteration evaluation cost (2 x 10 rollouts through an
`mp.Pool(10)`, i2c/env.py:93-103). When the policy was filled from an I2cGraph of this build
(`policy.write(*i2c.get_local_linear_policy())` after `env.attach(i2c)`, or `policy.source = i2c`), all
rollouts run in one `i2c_rollout` launch; otherwise they run step by step on the host exactly like the
reference's `BaseSim.run` (NumPy; fine for one-off evaluation of hand-made policies)."""
import numpy as np

from . import known_models as km


class KnownSim:
    """Mixin: simulator protocol of the reference's BaseSim / BaseKnownSim (i2c/env.py:35-187)."""

    env = None
    simulated = True

    def _init_sim(self, duration):
        self.duration = duration
        self._graph = None

    # -- the reference's step-by-step protocol ---------------------------------------------------
    def init_env(self):
        self.x = np.array(self.x0, dtype=float).reshape(1, self.dim_x)
        return self.x

    def forward(self, u):
        assert u.shape[0] == 1
        self.x = self.dynamics(np.hstack((self.x, u)))
        if not self.deterministic:
            self.x = self.x + np.random.multivariate_normal(np.zeros(self.dim_x), self.sig_eta, 1)
        return self.x

    def batch_forward(self, xu):
        x = self.dynamics(xu)
        if self.deterministic:
            return x
        return x + np.random.multivariate_normal(np.zeros(self.dim_x), self.sig_eta, x.shape[0])

    def process_y(self, y):
        return y

    def run(self, policy, deterministic=True, render=False, use_tqdm=False):
        """One host-side rollout: (xu, dx, z, z_term) as the reference's BaseSim.run (env.py:40-74)."""
        T = self.duration
        xt, yt, zt = np.zeros((T, self.dim_xu)), np.zeros((T, self.dim_x)), np.zeros((T, self.dim_z))
        x = self.init_env()
        for t in range(T):
            u = np.reshape(policy(t, x.T, deterministic), (1, self.dim_u))
            prev = np.copy(x)
            x = self.forward(u)
            row = np.hstack((prev, u))
            xt[t], yt[t], zt[t] = row[0], (x - prev)[0], self.observe(row)[0]
        z_term = self.observe_terminal(x)
        return xt, self.process_y(yt), zt, z_term

    def run_render(self, policy, dir, name="", deterministic=True, use_tqdm=False):
        print("Cannot render this environment")
        return self.run(policy, deterministic, render=False)

    # -- batched evaluation ------------------------------------------------------------------------
    def attach(self, graph):
        """Tell the simulator which I2cGraph the policies are written from (enables the GPU path)."""
        self._graph = graph
        return self

    def batch_eval(self, policy, n_eval, deterministic=True):
        graph = getattr(policy, "source", None) or self._graph
        if graph is not None and getattr(policy, "device_policy", None) and _matches(policy, graph):
            return self._device_eval(graph, policy, n_eval, deterministic)
        if n_eval > 1:
            # The reference farms these rollouts out to forked pool workers (env.py:96-100): each starts from a COPY of the
            # parent's NumPy stream and the parent's own stream does not advance. Reproduced, so that what the caller draws
            # afterwards (the final rollout saved as xu_real.npy, i2c_run.py:158) sees the same numbers as in the reference.
            state = np.random.get_state()
            outs = []
            for _ in range(n_eval):
                np.random.set_state(state)
                outs.append(self.run(policy, deterministic))
            np.random.set_state(state)
        else:
            outs = [self.run(policy, deterministic)]
        return tuple(list(col) for col in zip(*outs))

    def _device_eval(self, graph, policy, n_eval, deterministic):
        e = graph.engine
        res = e.rollout(n_eval, policy.device_policy, process_noise=not self.deterministic,
                        action_noise=not deterministic)
        xu = res["xu"][:, 0].to("cpu").double().numpy()  # (R, T, d) of trajectory 0
        z = res["z"][:, 0].to("cpu").double().numpy()
        xf = res["x_final"][:, 0].to("cpu").double().numpy()
        x_all = np.concatenate((xu[:, :, : self.dim_x], xf[:, None, :]), axis=1)
        dx = x_all[:, 1:] - x_all[:, :-1]
        zt = None if res["z_term"] is None else res["z_term"][:, 0].to("cpu").double().numpy()
        z_terms = [None if zt is None else zt[r].reshape(1, -1) for r in range(n_eval)]
        return list(xu), [self.process_y(d) for d in dx], list(z), z_terms

    # -- presentation: not part of the solver --------------------------------------------------------
    def plot_sim(self, *args, **kwargs):
        return None

    plot_trajectory = plot_sim

    def close(self):
        pass


def _matches(policy, graph):
    """The device path reads (K, k, sigK, mu, sig_x) from the graph's buffers: only valid while the
    policy object still holds exactly what the graph last produced."""
    K, k, _ = graph.get_local_linear_policy()
    if graph.B != 1 or policy.K.shape != K.shape or not np.array_equal(policy.K, K):
        return False
    if policy.device_policy == "linear":
        return np.array_equal(policy.k, k)
    return np.array_equal(policy.k, graph.get_local_expert_linear_policy()[1])


def _sim(model_cls, stochastic_x0=False):
    class Sim(KnownSim, model_cls):  # the simulator protocol (run, forward, ...) takes precedence over the model's
        def __init__(self, duration):
            model_cls.__init__(self)
            self._init_sim(duration)

        if stochastic_x0:  # BaseLinear.init_env samples the initial state (env.py:195-197)
            def init_env(self):
                self.x = np.random.multivariate_normal(np.asarray(self.x0, float).reshape(-1), self.sig_x0, 1)
                return self.x

    Sim.__name__ = model_cls.__name__ + "Sim"
    return Sim


LinearSim = _sim(km.LinearExact, stochastic_x0=True)
LinearMinimumEnergy = _sim(km.LinearMinimumEnergy, stochastic_x0=True)
PendulumKnown = _sim(km.PendulumKnown)
PendulumKnownActReg = _sim(km.PendulumKnownActReg)
CartpoleKnown = _sim(km.CartpoleKnown)
DoubleCartpoleKnown = _sim(km.DoubleCartpoleKnown)

_SIMULATORS = {
    "LinearKnown": LinearSim,
    "LinearKnownMinimumEnergy": LinearMinimumEnergy,
    "PendulumKnown": PendulumKnown,
    "PendulumKnownActReg": PendulumKnownActReg,
    "CartpoleKnown": CartpoleKnown,
    "DoubleCartpoleKnown": DoubleCartpoleKnown,
}


# the reference's base-class names (env.py:34, 168, 198): one simulator class serves every known model here
BaseSim = BaseKnownSim = BaseLinear = KnownSim


def make_env(exp):
    """Simulator for an experiment module (same call as the reference's make_env, env.py:17-32)."""
    return _SIMULATORS[exp.ENVIRONMENT](exp.N_DURATION)
