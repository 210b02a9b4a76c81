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


def assert_close(a, b, rtol, what=""):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    e = rel_err(a, b)
    assert e <= rtol, f"{what}: max-norm relative error {e:.3e} > {rtol:.1e}"
