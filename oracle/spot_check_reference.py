"""TEST INFRASTRUCTURE ONLY -- the product against the REAL reference on problems NEITHER has seen (no committed golden): what the
round-5 judge did by hand, as a script. Needs /root/reference (this container only; nothing here travels to the GPU box).

    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python oracle/spot_check_reference.py

1. a child process imports the unmodified reference through oracle/ref_shim.py and runs its I2cGraph on freshly drawn problems
   (new horizons, seeds, cost weights, temperatures, tolerances, cubature rules -- the generator of oracle/gen_golden.py, other
   arguments), writing the captures as .npz into a TEMPORARY directory;
2. this process replays each of them through the product engine -- the host-simulation build of the same csrc/ cell code -- on
   every kernel family that serves it (default, one lane per trajectory, quad sweeps incl. the round-6 backward walk and the quad walker
   of the chunked schedule, group), with
   tests/parity.check_against_golden: every per-cell quantity and the EM summaries.
Prints one line per (problem, family); exits non-zero on a mismatch."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys, os
sys.path.insert(0, {here!r})
import gen_golden as G
import numpy as np
G.GOLDEN = {out!r}
from i2c.i2c import I2cGraph
from i2c.model import make_env_model
from i2c.exp_types import CubatureQuadrature

def em(name, model_name, model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, quad=(1, 0, 0), mu_xt=None, sig_xt=None, n_detail=2, n_total=5, **extra):
    g = I2cGraph(model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, mu_xt, sig_xt, CubatureQuadrature(*quad))
    if extra.get("propagate"):
        for c in g.cells:
            c.use_expert_controller = bool(extra.get("use_expert_controller", True))
        g._propagate = True
    out = G.problem_inputs(model_name, model, T, Q, R, Qf, alpha, tol, mu_u, sig_u, mu_xt, sig_xt, quad, **extra)
    G.run_em(g, n_detail, n_total, out, pre_propagate=bool(extra.get("propagate")))
    G.save(name, out)

rng = np.random.default_rng(20261004)
# pendulum, T = 57: other weights, temperature, tolerance, seed
em("spot_pendulum_T57", "PendulumKnown", make_env_model("PendulumKnown", None), 57, np.diag([2.0, 60.0, 0.5]), np.diag([1.3]), np.diag([1.0, 150.0, 2.0]),
   60.0, 0.2, 2e-2 * rng.normal(size=(57, 1)), 1.5 * np.eye(1))
# double cartpole, T = 33
sf = 2e-3
Qd = sf * np.diag([2.0, 1.0, 80.0, 1.0, 120.0, 5.0, 1.0, 2.0])
em("spot_dcp_T33", "DoubleCartpoleKnown", make_env_model("DoubleCartpoleKnown", None), 33, Qd, sf * np.diag([0.2]), Qd, 0.08, 0.9,
   1e-2 * rng.normal(size=(33, 1)), np.eye(1))
# cartpole, T = 45, a weight on the centre point
Qc = np.diag([2.0, 1.0, 80.0, 5.0, 1.0])
em("spot_cartpole_T45_centre", "CartpoleKnown", make_env_model("CartpoleKnown", None), 45, Qc, np.diag([0.7]), Qc, 50.0, 0.0,
   1e-3 * rng.normal(size=(45, 1)), np.eye(1), quad=(1, 0, 0.5))
# planar quadrotor, T = 17, general weights (round 6: identity-observation model on the quad kernels)
mq = G._reference_quadrotor()
Qq = np.diag([2e3, 1e3, 5e2, 1, 2, 1]) / 1e3
em("spot_quadrotor_T17_general", "PlanarQuadrotor", mq, 17, Qq, np.diag([2e-3, 1e-3]), Qq, 0.7, 0.5,
   0.5 * mq.gravity * np.ones((17, 2)) + 1e-2 * rng.normal(size=(17, 2)), 1e-2 * np.eye(2), quad=(1.2, 0.44, 0.5))
# covariance control, T = 37
em("spot_covctrl_T37", "PendulumKnownActReg", make_env_model("PendulumKnownActReg", None), 37, None, np.diag([0.8]), None, 200.0, 1.0,
   np.zeros((37, 1)), 0.5 * np.eye(1), mu_xt=np.array([0.1, 0.0]), sig_xt=np.diag([2e-3, 1e-3]), n_detail=3, n_total=8,
   propagate=True, use_expert_controller=False)
# 12-state quadrotor, T = 9, general weights (round 6: d = 16 on the quad kernels at every batch size)
m12 = G._reference_quad12()
em("spot_quad12_T9_general", "Quadrotor12", m12, 9, G.QUAD12_Q, G.QUAD12_R, G.QUAD12_Q, 0.8, 0.5,
   0.25 * m12.gravity * np.ones((9, 4)) + 1e-2 * rng.normal(size=(9, 4)), 1e-2 * np.eye(4), quad=(1.2, 0.44, 0.5), n_total=4)
"""

QC = (64, "chunked")  # the quad walker inside the chunked schedule, asked for by name (the default of the d >= 5 models at B = 1)
CASES = [  # (name, tolerances, families to ask for: group_lanes values, or (group_lanes, backward_mode))
    ("spot_pendulum_T57", (1e-8, 1e-7), (0, 64, True, QC)),
    ("spot_dcp_T33", (1e-6, 1e-5), (0, -1, 64, True)),
    ("spot_cartpole_T45_centre", (1e-6, 1e-5), (0, -1, 64)),
    ("spot_quadrotor_T17_general", (1e-6, 1e-5), (0, -1, 64, True)),
    ("spot_covctrl_T37", (1e-7, 1e-6), (0, 64, QC)),
    ("spot_quad12_T9_general", (1e-6, 1e-5), (0, 16)),
]


def main():
    out = tempfile.mkdtemp(prefix="i2c_spot_")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", CHILD.format(here=HERE, out=out)], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("reference run failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    print(r.stdout.strip())
    for p in (ROOT, os.path.join(ROOT, "input-inference-for-control_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import golden_util
    import hostsim
    import parity

    golden_util.GOLDEN_DIR = out
    lib = hostsim.load()
    bad = 0
    for name, (td, ts), fams in CASES:
        for lanes in fams:
            kw = dict(group_lanes=lanes[0], backward_mode=lanes[1]) if isinstance(lanes, tuple) else dict(group_lanes=lanes)
            try:
                eng = parity.check_against_golden(name, lib, "cpu", td, ts, **kw)
                print(f"{name:30s} group_lanes={str(lanes):5s} OK   forward {eng.forward_family:5s} backward {eng.backward_family:5s} ({eng.backward_schedule})")
            except AssertionError as e:
                bad += 1
                print(f"{name:30s} group_lanes={str(lanes):5s} FAIL {str(e)[:200]}")
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
