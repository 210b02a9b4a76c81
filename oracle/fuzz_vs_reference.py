"""TEST INFRASTRUCTURE ONLY -- randomised check of the CPU oracle against the REAL reference, run in this container (the
reference cannot travel):

    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python oracle/fuzz_vs_reference.py [seed] [cases]

Random problems (model, inference rule, horizon, cost weights -- diagonal or coupled --, temperature, update tolerance,
feedback horizon tau, terminal cost on / off, terminal state prior, propagation, expert controller) through the reference's
I2cGraph and through oracle/i2c_numpy.py / i2c_linearize_numpy.py; prints the worst relative deviation of the marginals, the
controllers, alpha and the plan cost per case. The golden cases pin fixed combinations; this covers random ones. The result of
the last run is recorded in oracle/FUZZ_RESULT.txt."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402

ref_shim.install()

import numpy as np  # noqa: E402
from i2c.exp_types import CubatureQuadrature, GaussHermiteQuadrature, Linearize  # noqa: E402  (the reference)
from i2c.i2c import I2cGraph  # noqa: E402
from i2c.model import make_env_model  # noqa: E402

from oracle.i2c_linearize_numpy import I2cLinearizeOracle  # noqa: E402
from oracle.i2c_numpy import CubatureRule, GaussHermiteRule, I2cOracle  # noqa: E402
from oracle.models_numpy import make_model  # noqa: E402

# (environment, Q diag, R diag, alpha, sig_u): hyper-parameters of the reference's experiment files
MODELS = {
    "PendulumKnown": ([1, 100.0, 1], [2.0], 100.0, 2.0),
    "CartpoleKnown": ([1.0, 1.0, 100.0, 10.0, 1.0], [1.0], 80.0, 1.0),
    "DoubleCartpoleKnown": ([1e-3 * v for v in (1.0, 1.0, 100.0, 1.0, 100.0, 10.0, 1.0, 1.0)], [1e-4], 0.05, 1.0),
    "LinearKnown": ([10.0, 10.0], [1.0], 100.0, 100.0),
}
IDENTITY_TERMINAL = {"LinearKnown"}


def spd(rng, diag, coupled):
    d = np.asarray(diag, float) * 10.0 ** rng.uniform(-0.5, 0.5, size=len(diag))
    if not coupled or len(d) == 1:
        return np.diag(d)
    q, _ = np.linalg.qr(rng.normal(size=(len(d), len(d))))
    q = np.eye(len(d)) + 0.2 * (q - np.eye(len(d)))  # mild coupling: keeps the scale of every direction
    m = q @ np.diag(d) @ q.T
    return 0.5 * (m + m.T)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-9))


def one(rng, idx):
    name = list(MODELS)[rng.integers(len(MODELS))]
    qd, rd, alpha0, su = MODELS[name]
    kind = ["cubature", "cubature", "linearize", "gauss_hermite"][rng.integers(4)]
    if kind == "gauss_hermite" and name not in ("PendulumKnown", "LinearKnown"):
        kind = "cubature"  # (3^5 ... 3^7 grid points per transform in the reference's Python loops)
    T = int(rng.integers(2, 10))
    coupled = bool(rng.integers(2))
    Q, R = spd(rng, qd, coupled), spd(rng, rd, False)
    model = make_env_model(name, None)
    if name == "LinearKnown":
        model.sig_x0 = 1e-4 * np.eye(2)
        model.sig_eta = 1e-4 * np.eye(2)
    nzt = model.dim_z_term if hasattr(model, "dim_z_term") else len(qd)
    Qf = spd(rng, qd[:nzt] if len(qd) >= nzt else qd, coupled) if (kind == "linearize" or rng.integers(3) > 0) else None
    if Qf is not None and Qf.shape[0] != nzt:
        Qf = np.diag(np.diag(Q)[:nzt])
    xterm = rng.integers(4) == 0 and (kind != "linearize" or name in IDENTITY_TERMINAL)
    mu_xt = sig_xt = None
    if xterm:
        nx = model.dim_x
        mu_xt = 0.1 * rng.normal(size=nx)
        sig_xt = np.diag(10.0 ** rng.uniform(-3, -1, size=nx))
    alpha = alpha0 * 10 ** rng.uniform(-0.5, 0.5)
    tol = float(rng.choice([0.0, 0.5, 0.99, 1.0]))
    mu_u = 1e-2 * rng.normal(size=(T, model.dim_u))
    sig_u = su * 10 ** rng.uniform(-0.3, 0.3) * np.eye(model.dim_u)
    propagate = bool(rng.integers(2)) or xterm
    expert = bool(rng.integers(2))
    tau = int(rng.integers(0, T)) if rng.integers(2) else None
    rule = {"cubature": CubatureQuadrature(1, 0, 0), "linearize": Linearize(), "gauss_hermite": GaussHermiteQuadrature(3)}[kind]
    g = I2cGraph(model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, None if mu_xt is None else mu_xt.reshape(-1, 1), sig_xt, rule)
    for c in g.cells:
        c.use_expert_controller = expert
    g._propagate = propagate
    if tau is not None:
        g.tau = tau
    om = make_model(name, **({"noise": 1e-4} if name == "LinearKnown" else {}))
    cls = I2cLinearizeOracle if kind == "linearize" else I2cOracle
    orule = GaussHermiteRule(3) if kind == "gauss_hermite" else CubatureRule(1, 0, 0)
    o = cls(om, T, Q, R, Qf, alpha, tol, mu_u, sig_u, mu_xt, sig_xt, orule)
    o._propagate = propagate
    o.use_expert_controller = expert
    if tau is not None:
        o.tau = tau
    if propagate:
        g.propagate()
        o.propagate()
    worst, where = 0.0, ""
    for it in range(3):
        g.learn_msgs()
        o.learn_msgs()
        K, k, sigK = g.get_local_linear_policy()
        mu, sig = g.get_marginal_state_action_distribution()
        for what, a, b in (("mu", o.mu_xu0_m[0], mu), ("sig", o.sig_xu0_m[0], sig), ("K", o.K[0], K), ("k", o.k[0], k),
                           ("sigK", o.sigK[0], sigK), ("alpha", o.alpha[0], g.alpha), ("cost", o.costs_m[-1][0], g.costs_m[-1])):
            e = rel(a, b)
            if not np.isfinite(e):
                e = np.inf
            if e > worst:
                worst, where = e, f"it{it} {what}"
    desc = f"{name} {kind} T={T} coupled={int(coupled)} Qf={Qf is not None} xterm={int(bool(xterm))} prop={int(propagate)} expert={int(expert)} tau={tau} tol={tol}"
    print(f"{idx:3d} {desc}: worst {worst:.1e} at {where}", flush=True)
    return worst, desc


if __name__ == "__main__":
    import logging

    logging.disable(logging.CRITICAL)
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    res = []
    for i in range(n):
        try:
            res.append(one(rng, i))
        except Exception as e:  # (so far always the reference: a covariance that is not positive definite, quadrature.py:18)
            print(f"{i:3d} reference / oracle raised {type(e).__name__}: {str(e)[:120]}", flush=True)
    w = max(res) if res else (0.0, "")
    print(f"seed {seed}: {len(res)} of {n} cases compared, worst {w[0]:.2e} ({w[1]})")
