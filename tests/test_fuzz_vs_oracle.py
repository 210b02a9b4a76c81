"""Randomised differential test: random problems (model, inference rule, horizon, batch, cost weights -- diagonal or coupled --,
temperature, update tolerance, feedback horizon, propagation, expert controller) through the batched engine and through the CPU
oracle. Complements tests/test_feature_matrix.py (one option at a time) with combinations; the seeds are fixed."""
import json

import numpy as np
import pytest

import hostsim
import parity
from golden_util import Case, load_case, oracle_from_case

BASES = ["em_pendulum_T200", "lin_pendulum_T100", "em_linear_T60", "lin_linear_T60", "em_cartpole_T100", "lin_cartpole_T100",
         "em_dcp_T60", "em_quadrotor_T20", "em_quad12_T20", "lin_quad12_T20", "gh3_pendulum_T40"]


def _spd(rng, n, lo, hi, coupled):
    d = 10.0 ** rng.uniform(np.log10(lo), np.log10(hi), size=n)
    if not coupled:
        return np.diag(d)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    m = q @ np.diag(d) @ q.T
    return 0.5 * (m + m.T)


def random_case(rng, bases=BASES):
    name = bases[rng.integers(len(bases))]
    g = load_case(name)
    meta = dict(g.meta)
    T, B = int(rng.integers(2, 16)), int(rng.integers(1, 5))
    d = {k: g[k] for k in g}
    nu = g["R"].shape[0]
    coupled = bool(rng.integers(2))
    if "Q" in g:
        d["Q"] = _spd(rng, g["Q"].shape[0], 0.1, 100, coupled) * (np.abs(g["Q"]).max() / 100)
    d["R"] = _spd(rng, nu, 0.1, 10, coupled and nu > 1) * np.abs(g["R"]).max()
    if "Qf" in g and rng.integers(3) > 0:
        d["Qf"] = _spd(rng, g["Qf"].shape[0], 0.1, 100, coupled) * (np.abs(g["Qf"]).max() / 100)
    elif "Qf" in g and meta.get("inference") != "linearize":  # (Linearize without a terminal cost: the reference fails, see
        d.pop("Qf")                                            #  test_feature_matrix.py)
    meta.update(T=T, alpha=float(meta["alpha"] * 10 ** rng.uniform(-1, 1)), tol=float(rng.choice([0.0, 0.5, 0.99, 1.0])))
    if rng.integers(2):
        meta["propagate"] = True
    if rng.integers(2):
        meta["use_expert_controller"] = bool(rng.integers(2))
    if rng.integers(2):
        meta["tau"] = int(rng.integers(0, T + 1))
    d["meta"] = np.array(json.dumps(meta))
    d["mu_u"] = g["mu_u"][:T] * (1 + 0.3 * rng.normal(size=(T, nu))) + 1e-3 * rng.normal(size=(T, nu))
    d["sig_u"] = g["sig_u"] * 10 ** rng.uniform(-0.5, 0.5)
    return name, Case(d), B, int(rng.integers(1 << 30)), float(10 ** rng.uniform(-3, -1))


# kernel families a model can be asked for (cubature rule): 0 = its default, -1 = one lane per trajectory, True = the group kernels,
# 64 = the matrix-instruction family (wave kernels where they exist, the quad forward kernel otherwise), LANES_QUAD = the quad
# kernels of the model that also has wave kernels (forward and backward sweep)
# (64, "chunked") = the quad walker inside the chunked schedule (d <= 8; horizons below 8 cells fall back to the fused quad walk)
QC = (64, "chunked")
FAMILIES = {"em_pendulum_T200": (0, 64, True, QC), "em_linear_T60": (0, 64, QC), "em_cartpole_T100": (0, -1, 64, True, QC), "em_dcp_T60": (0, -1, 64, True, QC),
            "em_quadrotor_T20": (0, -1, 64, True, QC), "em_quad12_T20": (0, 16, 64, parity.pkg._native.LANES_QUAD)}


# cubature rules for the random-weights variant (round 6): unit; lam != 0 with W = 1; weights that do not sum to one (W = 0.99, 0.8975).
# (W > 1 is not usable: the reference's own covariances lose positive definiteness)
WEIGHTS = [(1.0, 0.0, 0.0), (1.2, 0.44, 0.5), (1.0, 0.0, 0.5), (1.0, -0.01, 0.0), (1.05, 0.0, 0.3)]


def run_random_case(lib, device, seed, tol, random_family=False, random_weights=False):
    rng = np.random.default_rng(seed)
    name, case, B, seed_b, x0_scale = random_case(rng, sorted(FAMILIES) if random_family else BASES)
    x0, mu_u = parity.batched_inputs(case, B, seed=seed_b, x0_scale=x0_scale)
    kw, kw_skip = {}, ()
    fam = FAMILIES.get(name, (0,))
    if random_weights and case.meta.get("inference", "cubature") == "cubature":  # (a Linearize / Gauss-Hermite graph has no such parameter)
        quad = WEIGHTS[np.random.default_rng(seed + 104729).integers(len(WEIGHTS))]
        case = Case({**case, "meta": np.array(json.dumps({**case.meta, "quad": list(quad)}))})
        if name == "em_quad12_T20" and quad != WEIGHTS[0]:
            fam = (0, 16)  # d = 16: general weights are the group kernels' (the wave / quad forms refuse them)
        if abs(2.0 - quad[0] ** 2 + quad[1] - 1.0) > 1e-12:
            # W != 1: the rule's (W - W^2) m m^T terms cancel against the covariances (|m|^2 / sigma^2 ~ 1e4 on the quadrotors): EVERY
            # family and the oracle agree to ~1e-9 on covariances and ~1e-5 on the small gains there (seed 307), unit rules to 1e-10
            tol = max(tol, 3e-5)
            kw_skip = ("K", "k") if name in ("em_quadrotor_T20", "em_quad12_T20") else ()  # (their gains are ~1e-3 and carry that noise at 1e-4)
    if random_family and name in FAMILIES:
        kw["group_lanes"] = fam[np.random.default_rng(seed + 7919).integers(len(fam))]
        if isinstance(kw["group_lanes"], tuple):
            kw["group_lanes"], kw["backward_mode"] = kw["group_lanes"]
    eng = parity.engine_from_case(case, lib, device, x0=x0, mu_u=mu_u, **kw)
    o = oracle_from_case(Case({**case, "mu_u": mu_u}), x0=x0)
    if case.meta.get("propagate"):
        eng.propagate()
        o.propagate()
    amplifying, worst = bool(kw_skip) or tol >= 3e-5, 0.0  # (set with the W != 1 tolerance above)
    for it in range(3):
        eng.learn_msgs()
        try:
            o.learn_msgs()
        except np.linalg.LinAlgError:
            # the REFERENCE's algorithm breaks down on this draw (weights that do not sum to one; a 12-state Linearize() problem that
            # diverges over a longer horizon): nothing to compare -- but on a unit rule the kernels must have flagged a trajectory too
            assert random_weights or eng.failures(), "the oracle lost positive definiteness on a unit-rule problem the kernels call healthy"
            return
        mu, sig = eng.marginal_state_action()
        K, k, sigK = eng.local_linear_policy()
        for what, a, b in (("mu", mu, o.mu_xu0_m), ("sig", sig, o.sig_xu0_m), ("K", K, o.K), ("k", k, o.k), ("sigK", sigK, o.sigK),
                           ("alpha", eng.alpha, o.alpha), ("cost", eng.costs_m[-1], o.costs_m[-1])):
            if what in kw_skip:
                continue
            a, b = parity.np_(a), np.asarray(b, float)
            err = np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-9)  # (the floor: gains that are zero up to rounding)
            assert np.isfinite(err) and err <= tol, f"seed {seed} ({name}, {case.meta}) it{it} {what}: {err:.2e}"
            worst = max(worst, err) if what in ("mu", "sig") else worst
        if amplifying and worst > 1e-7:
            # W != 1 with an unclamped temperature update: some draws amplify ANY deviation a hundred- to a thousandfold per EM iteration
            # (seed 111107, double cartpole T = 12: 6e-11 -> 1e-6 -> 5e-4 on the quad kernels, 2e-10 -> 5e-6 -> 3e-3 on the lane
            # kernels, every family against the same oracle): once the posterior has left the 1e-7 neighbourhood the next
            # iteration compares two different problems
            break
    assert eng.failures() == [], f"seed {seed} ({name})"


@pytest.mark.parametrize("seed", range(100, 130))
def test_hostsim_random_problems_vs_oracle(seed):
    run_random_case(hostsim.load(), "cpu", seed, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 130))
def test_hip_random_problems_vs_oracle(seed):
    run_random_case(parity.pkg.load_library(), "cuda", seed, 1e-5)


# the same random problems with a random kernel family asked for (round 4: four families per model, two of them new)
@pytest.mark.parametrize("seed", range(200, 240))
def test_hostsim_random_problems_random_family_vs_oracle(seed):
    run_random_case(hostsim.load(), "cpu", seed, 1e-6, random_family=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(200, 240))
def test_hip_random_problems_random_family_vs_oracle(seed):
    run_random_case(parity.pkg.load_library(), "cuda", seed, 1e-5, random_family=True)


# ... and with a random cubature rule on top (round 6: a general-weights bug of the quad forward kernel -- the unweighted process
# noise -- had survived because every general-weights test used a rule whose weights happen to sum to one)
@pytest.mark.parametrize("seed", range(300, 330))
def test_hostsim_random_problems_random_family_random_weights_vs_oracle(seed):
    run_random_case(hostsim.load(), "cpu", seed, 1e-6, random_family=True, random_weights=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(300, 360))
def test_hip_random_problems_random_family_random_weights_vs_oracle(seed):
    run_random_case(parity.pkg.load_library(), "cuda", seed, 1e-5, random_family=True, random_weights=True)
