"""input-inference-for-control_amd: MI355X-native batched Gaussian i2c solver.

    import importlib
    i2c_amd = importlib.import_module("input-inference-for-control_amd")
    solver = i2c_amd.BatchedI2c(model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, ...)

The sub-package ``i2c`` mirrors the reference's public names (``i2c.i2c.I2cGraph``,
``i2c.exp_types``, ``i2c.model.make_env_model`` ...): put this directory on ``sys.path`` and
the reference's scripts import the MI355X build instead.
"""
from ._native import load_library, MODEL_IDS  # noqa: F401
from . import dist, engine  # noqa: F401
from .engine import BatchedI2c, I2cNumericalError  # noqa: F401
