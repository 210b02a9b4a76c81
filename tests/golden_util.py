"""Helpers shared by the parity tests: load a golden case (tests/golden/*.npz, captured from
the real reference by oracle/gen_golden.py) and build the CPU oracle for the same problem."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case(dict):
    @property
    def meta(self):
        return json.loads(str(self["meta"]))

    def iters(self):
        return sorted({int(k.split("/")[0][2:]) for k in self if k.startswith("it")})

    def at(self, it, key):
        return self[f"it{it}/{key}"]

    def has(self, it, key):
        return f"it{it}/{key}" in self


def load_case(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as f:
        return Case({k: f[k] for k in f.files})


def oracle_model(case):
    from oracle.models_numpy import make_model

    meta = case.meta
    kw = {}
    if meta["model"] == "LinearKnown" and "noise" in meta:
        kw["noise"] = meta["noise"]
    if meta["model"] == "LinearKnown" and "goal" in meta:
        kw["goal"] = meta["goal"]
    m = make_model(meta["model"], **kw)
    return m


def oracle_from_case(case, **over):
    from oracle.i2c_numpy import CubatureRule, GaussHermiteRule, I2cOracle

    meta = case.meta
    model = oracle_model(case)
    cls = I2cOracle
    if meta.get("inference") == "linearize":
        from oracle.i2c_linearize_numpy import I2cLinearizeOracle as cls
    o = cls(
        model,
        meta["T"],
        case.get("Q"),
        case["R"],
        case.get("Qf"),
        meta["alpha"],
        meta["tol"],
        case["mu_u"],
        case["sig_u"],
        case.get("mu_x_term"),
        case.get("sig_x_term"),
        GaussHermiteRule(meta["gh_degree"]) if meta.get("inference") == "gauss_hermite" else CubatureRule(*meta["quad"]),
        **over,
    )
    if meta.get("propagate"):
        o._propagate = True
    if "use_expert_controller" in meta:
        o.use_expert_controller = bool(meta["use_expert_controller"])
    if "tau" in meta:
        o.tau = int(meta["tau"])
    return o


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


# Element-wise mode (round-4 review, weak #1): a max-norm over a whole (T, ...) array can hide a wrong SMALL entry. Every entry
# must satisfy |a - b| <= rtol |b| + atol, rtol = the north star's own numbers (1e-5 on means, 1e-4 on covariances / gains) and
# atol = atol_rel * max|b|. The absolute floor is the rounding noise that the LARGE entries of an array leave on its small ones,
# in the reference as much as here (an entry that is 1e-9 of the array's maximum and was formed by sums with cancellation, or by
# a solve against an ill-conditioned covariance, carries ~1e-16 * cond / 1e-9 relative noise): it scales with the conditioning
# of the case, which is what the case's max-norm tolerance `tol` reflects -- parity.close() uses atol_rel = max(1e-12, tol / 100):
# 1e-11 for the pendulum goldens (tol 1e-9), 1e-9 for the quadrotors (tol 1e-7). The review's 1e-12 holds for none of the d >= 7
# models: the reference's own J_dyn of the planar quadrotor differs from the oracle's by 7e-10 of the array maximum.
ATOL_REL = 1e-10
ELEM_RTOL_MEAN = 1e-5
ELEM_RTOL_COV = 1e-4


def elem_rtol_for(key):
    """North-star tolerance class of a quantity by its name: means (mu_*, k, alpha, costs) 1e-5, covariances and gains 1e-4."""
    k = key.split("/")[-1].split()[-1]
    return ELEM_RTOL_MEAN if (k.startswith("mu_") or k in ("k", "alpha", "cost")) else ELEM_RTOL_COV


def elementwise_excess(a, b, rtol, atol_rel=ATOL_REL):
    """max over entries of |a - b| / (rtol |b| + atol): <= 1 passes. Returns (excess, flat index of the worst entry)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0, 0
    bound = rtol * np.abs(b) + atol_rel * np.max(np.abs(b)) + 1e-300
    ratio = np.abs(a - b) / bound
    i = int(np.argmax(ratio))
    return float(ratio.reshape(-1)[i]), i


def assert_close(a, b, rtol, what="", elem_rtol=None, atol_rel=ATOL_REL):
    """Max-norm relative check at `rtol`; with `elem_rtol` ALSO the element-wise check |a - b| <= elem_rtol |b| + atol."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    e = rel_err(a, b)
    assert e <= rtol, f"{what}: max-norm relative error {e:.3e} > {rtol:.1e}"
    if elem_rtol is not None:
        x, i = elementwise_excess(a, b, elem_rtol, atol_rel)
        idx = np.unravel_index(i, a.shape) if a.size else ()
        assert x <= 1.0, (f"{what}: element {idx}: |{a.reshape(-1)[i] if a.size else 0:.6e} - {b.reshape(-1)[i] if b.size else 0:.6e}| exceeds "
                          f"{elem_rtol:.0e} |b| + {atol_rel:.0e} max|b| by {x:.2f}x")


def assert_close_per_cell(a, b, rtol, what="", axis=0, floor_rel=1e-7):
    """Max-norm relative error of every cell (slice along `axis`) on its own: a gain K that is wrong in a cell where it is small
    does not hide behind the cells where it is large. Cells whose reference is below floor_rel * max|b| are compared absolutely,
    against rtol * floor_rel * max|b| (the terminal cell's K is exactly 0 here and rounding noise in the reference and the oracle -- up to
    1e-12 of the array maximum in the batched cartpole runs; covariance control has such cells in mid-horizon)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    a2 = np.moveaxis(a, axis, 0).reshape(a.shape[axis], -1)
    b2 = np.moveaxis(b, axis, 0).reshape(b.shape[axis], -1)
    top = np.max(np.abs(b2)) + 1e-300
    scale = np.maximum(np.max(np.abs(b2), axis=1), floor_rel * top)
    err = np.max(np.abs(a2 - b2), axis=1) / scale
    t = int(np.argmax(err))
    assert err[t] <= rtol, f"{what}: cell {t}: max-norm relative error {err[t]:.3e} > {rtol:.1e} (cell scale {scale[t]:.3e}, array max {top:.3e})"
