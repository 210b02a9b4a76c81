"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Makes the *unmodified* reference at /root/reference importable in THIS container so that
``oracle/gen_golden.py`` can capture golden input/output vectors and so that
``oracle/i2c_numpy.py`` (the CPU restatement) can be pinned against the real thing.
The reference itself never travels to the GPU box; only the vectors in ``tests/golden`` do.

What is shimmed: five absent third-party modules are replaced by
inert stand-ins, and four NumPy aliases that were removed in NumPy >= 1.24 / 2.0 are put
back. None of them is on the cubature hot path (tikzplotlib = plotting, autograd /
numdifftools = Jacobians of the Linearize path, gym = unused import).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("I2C_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "i2c"))


def _central_difference_jacobian(fun, argnum=0, eps=1e-6):
    """Stand-in for autograd.jacobian (Linearize path only; autograd is not installed here).

    Complex-step differentiation, d f / d x_j = Im f(x + i h e_j) / h with h = 1e-30: exact to rounding for the
    reference's dynamics, which are analytic NumPy code (np.clip on a complex argument compares real parts and so
    has derivative 1 inside the limits and 0 outside, as autograd's clip rule). Falls back to central differences
    (error ~1e-10) for functions that reject complex input."""
    import numpy as np

    def jac(*args):
        x = np.array(args[argnum], dtype=float)
        f0 = np.asarray(fun(*args))
        out = np.zeros(f0.shape + x.shape)
        it = np.nditer(x, flags=["multi_index"])
        for _ in it:
            idx = it.multi_index
            col = None
            try:
                xc = x.astype(complex)
                xc[idx] += 1e-30j
                ac = list(args)
                ac[argnum] = xc
                with np.errstate(all="ignore"):
                    fc = np.asarray(fun(*ac))
                if np.iscomplexobj(fc) and np.all(np.isfinite(fc)):
                    col = fc.imag / 1e-30
            except Exception:  # noqa: BLE001 - any failure means "not complex-safe"
                col = None
            if col is None:
                xp = x.copy()
                xm = x.copy()
                xp[idx] += eps
                xm[idx] -= eps
                ap = list(args)
                am = list(args)
                ap[argnum] = xp
                am[argnum] = xm
                col = (np.asarray(fun(*ap)) - np.asarray(fun(*am))) / (2 * eps)
            out[(Ellipsis,) + idx] = col
        return out

    return jac


def install():
    """Install stubs + aliases and put the reference on sys.path. Idempotent."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # do not litter the read-only reference with __pycache__
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    os.environ.setdefault("MPLBACKEND", "Agg")
    import numpy as np

    for name, val in (("NINF", -np.inf), ("Inf", np.inf), ("float", float)):
        if not hasattr(np, name):
            setattr(np, name, val)
    if not hasattr(np, "asscalar"):
        np.asscalar = lambda a: np.asarray(a).item()

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("tikzplotlib", save=lambda *a, **k: None)
    stub("matplotlib2tikz", save=lambda *a, **k: None)
    stub("gym")
    ag = stub("autograd", jacobian=_central_difference_jacobian)
    ag.numpy = np
    sys.modules["autograd.numpy"] = np

    class _LazyJacobian:  # numdifftools.Jacobian stand-in (class-body use at env_def.py:405)
        def __init__(self, fun, *a, **k):
            self._j = _central_difference_jacobian(fun)

        def __call__(self, *args):
            return self._j(*args)

    stub("numdifftools", Jacobian=_LazyJacobian)

    for p in (REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "scripts")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import matplotlib

    matplotlib.use("Agg")
