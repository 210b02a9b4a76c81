"""Module-level dynamics functions under the reference's names (i2c/env_autograd.py:5, 25, 60: `<model>_dynamics(xu)` on rows of
[x | u], and `<model>_dydxu`). The reference differentiates them with autograd; here they are the NumPy dynamics of the known
models (known_models.py -- the host-side twins of the device functors the solver evaluates) and their Jacobians by central
differences (KnownModel.dydxu). `<model>_dydxu(xu)` returns the reference's layout for a batch of rows: (N, dim_x, N, dim_xu)
is what autograd's jacobian produces for (N, dim_xu) -> (N, dim_x); callers index [0, :, 0, :] (env_def.py dydxu methods), which
this module keeps."""
import numpy as np

from .known_models import CartpoleKnown, DoubleCartpoleKnown, PendulumKnown

_MODELS = {}


def _model(cls):
    if cls not in _MODELS:
        _MODELS[cls] = cls()
    return _MODELS[cls]


def _dydxu(cls, xu):
    m = _model(cls)
    xu = np.asarray(xu, float).reshape(-1, m.dim_xu)
    n = xu.shape[0]
    out = np.zeros((n, m.dim_x, n, m.dim_xu))
    for i in range(n):
        out[i, :, i, :] = m.dydxu(xu[i:i + 1])
    return out


def pendulum_dynamics(xu):
    return _model(PendulumKnown).dynamics(np.asarray(xu, float))


def cartpole_dynamics(xu):
    return _model(CartpoleKnown).dynamics(np.asarray(xu, float))


def double_cartpole_dynamics(xu):
    return _model(DoubleCartpoleKnown).dynamics(np.asarray(xu, float))


def pendulum_dydxu(xu):
    return _dydxu(PendulumKnown, xu)


def cartpole_dydxu(xu):
    return _dydxu(CartpoleKnown, xu)


def double_cartpole_dydxu(xu):
    return _dydxu(DoubleCartpoleKnown, xu)
