"""CPU-only check of the KERNEL MATH: csrc/ compiled as a host simulation (tests/hostsim.py)
and driven through the same C ABI + engine as on the GPU, against the golden vectors of the
real reference and against the batched CPU oracle. The GPU twins of these tests are in
tests/test_hip_parity.py."""
import numpy as np
import torch
import pytest

import hostsim
import parity
from golden_util import assert_close, load_case


@pytest.fixture(scope="module")
def lib():
    return hostsim.load()


GOLDEN = [
    ("em_pendulum_T200", 1e-9, 1e-8),
    ("em_pendulum_T40_quad_general", 1e-9, 1e-8),
    ("em_dcp_T60", 1e-7, 1e-6),
    ("em_cartpole_T100", 1e-7, 1e-6),
    ("em_linear_T60", 1e-9, 1e-8),
    ("em_covctrl_T100", 1e-8, 1e-7),
    ("em_covctrl_qf_T40", 1e-8, 1e-7),        # covariance control + terminal cost + expert controller
    ("em_pendulum_T50_propagate", 1e-9, 1e-8),
    ("em_pendulum_T30_tau7", 1e-9, 1e-8),     # feedback horizon in the middle (i2c.py:1210-1213)
    ("em_quadrotor_T20", 1e-7, 1e-6),
    ("em_quad12_T20", 1e-7, 1e-6),            # group kernels only (d = 16)
    ("em_quad12_T12_propagate", 1e-7, 1e-6),
    ("em_quad12_nondiag_T12", 1e-7, 1e-6),    # non-diagonal Q, R, Qf (i2c.py:781-789), with propagation
    ("em_dcp_nondiag_T30", 1e-7, 1e-6),       # the same on a trigonometric observation (hybrid: group forward, lane backward)
]


@pytest.mark.parametrize("name,tol_d,tol_s", GOLDEN)
def test_hostsim_vs_reference_golden(lib, name, tol_d, tol_s):
    parity.check_against_golden(name, lib, "cpu", tol_d, tol_s)


# The GROUP kernels (csrc/i2c_group.hpp: G lanes of a wavefront per trajectory, blocks row-distributed, exchange through
# LDS) on the models that also have one-lane-per-trajectory kernels: the same golden vectors. The host simulation runs
# the G lanes of a group as G threads with a barrier where the device has its LDS fence.
GROUP_GOLDEN = [
    ("em_pendulum_T200", 1e-9, 1e-8, 8),              # G = 4
    ("em_pendulum_T40_quad_general", 1e-9, 1e-8, None),  # weights that do not sum to 1
    ("em_dcp_T60", 1e-7, 1e-6, None),                 # G = 16, trigonometric observation
    ("em_cartpole_T100", 1e-7, 1e-6, None),           # G = 8
    ("em_linear_T60", 1e-9, 1e-8, None),
    ("em_pendulum_T50_propagate", 1e-9, 1e-8, None),  # propagation with the expert controller
    ("em_quadrotor_T20", 1e-7, 1e-6, None),           # G = 8, identity observation
    ("em_covctrl_T100", 1e-8, 1e-7, 8),               # covariance control: tempered terminal prior, propagation, KL
    ("em_covctrl_qf_T40", 1e-8, 1e-7, None),
    ("em_dcp_nondiag_T30", 1e-7, 1e-6, None),         # non-diagonal weights: g_cost_full on nz = 9
]


@pytest.mark.parametrize("name,tol_d,tol_s,n_iters", GROUP_GOLDEN)
def test_hostsim_group_kernels_vs_reference_golden(lib, name, tol_d, tol_s, n_iters):
    parity.check_against_golden(name, lib, "cpu", tol_d, tol_s, n_iters=n_iters, group_lanes=True)


# The d >= 7 models run their FORWARD sweep on the group kernels by default at small batches (the hybrid default), so the
# goldens above no longer reach their one-lane forward kernels: group_lanes = -1 forces one lane per trajectory everywhere.
LANE_GOLDEN = [
    ("em_dcp_T60", 1e-7, 1e-6),
    ("em_dcp_nondiag_T30", 1e-7, 1e-6),
    ("em_quadrotor_T20", 1e-7, 1e-6),
]


# The 12-state quadrotor has two multi-lane forms: its DEFAULT is the wave kernels (csrc/i2c_wave.hpp: one wavefront per
# trajectory, 16 x 16 blocks in the fp64 matrix-instruction layout; the goldens in GOLDEN above run them), group_lanes = 16 asks
# for the group kernels. The host simulation runs the 64 lanes as 64 threads and emulates the matrix / cross-lane instructions.
QUAD12 = ["em_quad12_T20", "em_quad12_T12_propagate", "em_quad12_nondiag_T12", "em_quad12_covctrl_T12"]  # (the last: covariance control)


@pytest.mark.parametrize("name", QUAD12)
@pytest.mark.parametrize("lanes", [16, 64])
def test_hostsim_quad12_both_families_vs_reference_golden(lib, name, lanes):
    eng = parity.check_against_golden(name, lib, "cpu", 1e-7, 1e-6, group_lanes=lanes)
    assert eng.forward_family == eng.backward_family == {16: "group", 64: "wave"}[lanes]


@pytest.mark.parametrize("mode", ["fused", "two_pass"])
def test_hostsim_wave_backward_schedules_vs_reference_golden(lib, mode):
    """The wave family's two backward schedules -- the fused walk, and the sequential nx x nx scan + one wave per (t, b) cell +
    reduction (on request only: measured slower on the device, see Impl::schedule) -- against the same golden run."""
    eng = parity.check_against_golden("em_quad12_T20", lib, "cpu", 1e-7, 1e-6, n_iters=4, group_lanes=64, backward_mode=mode)
    assert eng.backward_schedule == mode and eng.backward_family == "wave"


def test_hostsim_wave_kernels_are_the_quad12_default(lib):
    from golden_util import load_case

    eng = parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu")
    assert (eng.forward_family, eng.backward_family, eng.backward_schedule) == ("wave", "wave", "fused")
    # round 5: the closed-loop propagation and the state estimator's step on the quad kernels (propagate_quad_body, ckf_quad_body)
    assert eng.kernel_family("propagate") == "quad" and eng.kernel_family("filter") == "quad"
    grp = parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu", group_lanes=16)
    assert grp.kernel_family("filter") == "group" and grp.kernel_family("propagate") == "group"
    general = parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu", quad=(1.2, 0.44, 0.5))
    # weights with lam != 0 are not covered by the wave form: the quad kernels take over at every batch size (round 6; it was the group kernels)
    assert (general.forward_family, general.backward_family, general.kernel_family("propagate")) == ("quad", "quad", "quad")
    with pytest.raises(RuntimeError, match="-2"):
        parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu", quad=(1.2, 0.44, 0.5), group_lanes=64).forward_sweep()


# The QUAD forward kernel (csrc/i2c_quad.hpp: four trajectories per wavefront, 4 x 4 blocks on the small fp64 matrix instruction,
# one sigma point per lane through the model functors -- general observation functions included) on every d <= 8 model, with the
# lane kernels' backward schedules behind it: the same golden vectors. group_lanes = 64 asks for it. The host simulation runs the
# 64 lanes of a wave as 64 threads and emulates the matrix / DPP instructions.
QUAD_GOLDEN = [
    ("em_pendulum_T200", 1e-9, 1e-8, 8),       # 1 block everywhere; trigonometric observation; terminal update through sin / cos
    ("em_dcp_T60", 1e-7, 1e-6, None),          # d = 7 (2 blocks), nz = 9 (3 blocks, the last one a scalar pivot)
    ("em_dcp_nondiag_T30", 1e-7, 1e-6, None),  # non-diagonal weights (the forward sweep only sees sig_xi = alpha inv(QR))
    ("em_cartpole_T100", 1e-7, 1e-6, None),    # nx = 4: the pdf-ratio solve has no spare column, the innovation none at the terminal update
    ("em_linear_T60", 1e-9, 1e-8, None),       # identity observation, d = 3
    ("em_quadrotor_T20", 1e-7, 1e-6, None),    # identity observation, d = 8: sixteen points, no centre evaluation
    ("em_pendulum_T30_tau7", 1e-9, 1e-8, None),  # feed-forward and feedback cells in one sweep
    ("em_covctrl_T100", 1e-8, 1e-7, 8),        # nz = 1 (only the action is observed), no terminal observation; covariance control behind it
    ("em_covctrl_qf_T40", 1e-8, 1e-7, None),
    ("em_pendulum_T40_quad_general", 1e-9, 1e-8, None),  # CubatureQuadrature(1.2, 0.44, 0.5): a weight on the centre, weights that do not
                                                        # sum to one -- the GENERAL variant of the quad kernel (round 5)
]


@pytest.mark.parametrize("name,tol_d,tol_s,n_iters", QUAD_GOLDEN)
def test_hostsim_quad_forward_vs_reference_golden(lib, name, tol_d, tol_s, n_iters):
    """(LANES_QUAD asks for the quad FORWARD sweep only; with a lane schedule asked for by name the lane backward sweep runs behind
    it: the default pair of the d >= 5 models at the BASELINE batch sizes)"""
    eng = parity.check_against_golden(name, lib, "cpu", tol_d, tol_s, n_iters=n_iters, group_lanes=parity.pkg._native.LANES_QUAD, backward_mode="chunked")
    assert eng.forward_family == "quad" and eng.backward_family == "lane" and eng.backward_schedule == "chunked"


# The quad WALKER of the chunked schedule (backward_quad8_body<CHUNK>): compose + stitch on the lane kernels, then four trajectories
# per wavefront and chunk walk their cells from the stitched boundary state; cost sums per chunk into the common reduction.
# group_lanes = 64 with "chunked" asks for it; the default of the d >= 5 models up to a few hundred trajectories.
@pytest.mark.parametrize("name,tol_d,tol_s,n_iters", QUAD_GOLDEN)
def test_hostsim_quad_chunk_walk_vs_reference_golden(lib, name, tol_d, tol_s, n_iters):
    eng = parity.check_against_golden(name, lib, "cpu", tol_d, tol_s, n_iters=n_iters, group_lanes=64, backward_mode="chunked")
    assert (eng.forward_family, eng.backward_family, eng.backward_schedule) == ("quad", "quad", "chunked")
    assert eng.work is not None


# The QUAD backward sweep of the d <= 8 models (round 6, backward_quad8_body): the fused walk of four trajectories per wavefront --
# posterior observation through sigma points, controller by a blocked back substitution with the joint's factor, the tempered
# terminal prior of covariance control and the terminal observation at the end of the chain -- on the same goldens, every per-cell
# quantity (posterior, controller, observation moments, smoothed state, terminal moments) and the EM summaries.
@pytest.mark.parametrize("name,tol_d,tol_s,n_iters", QUAD_GOLDEN)
def test_hostsim_quad_backward_vs_reference_golden(lib, name, tol_d, tol_s, n_iters):
    eng = parity.check_against_golden(name, lib, "cpu", tol_d, tol_s, n_iters=n_iters, group_lanes=64)
    assert (eng.forward_family, eng.backward_family, eng.backward_schedule) == ("quad", "quad", "fused")
    assert eng.work is None  # (one pass: no chunk workspace)


def test_hostsim_quad_backward_resolver(lib):
    """Which backward sweep a problem gets (i2c_kernel_family / i2c_backward_schedule, the one resolver). group_lanes = 64: the quad
    fused walk with the schedule left open or "fused", the quad walker inside the chunked schedule with "chunked", the lane kernels
    with "two_pass". LANES_QUAD asks for the forward sweep only: the backward sweep resolves as the default does -- the quad walker
    inside the model's measured window (a few hundred trajectories: the double cartpole's is 1 ... 256), the lane kernels beyond it
    and whenever a lane schedule is asked for by name."""
    g = load_case("em_dcp_T60")
    fam = lambda **kw: (lambda e: (e.forward_family, e.backward_family, e.backward_schedule))(parity.engine_from_case(g, lib, "cpu", **kw))  # noqa: E731
    Q = parity.pkg._native.LANES_QUAD
    assert fam(group_lanes=64) == fam(group_lanes=64, backward_mode="fused") == ("quad", "quad", "fused")
    assert fam(group_lanes=64, backward_mode="chunked") == ("quad", "quad", "chunked")
    assert fam(group_lanes=64, backward_mode="two_pass") == ("quad", "lane", "two_pass")
    assert fam() == fam(group_lanes=Q) == ("quad", "quad", "chunked")  # (B = 1: inside the default window)
    assert fam(group_lanes=Q, backward_mode="chunked") == fam(backward_mode="chunked") == ("quad", "lane", "chunked")
    assert fam(group_lanes=Q, backward_mode="fused") == ("quad", "lane", "fused")
    assert fam(group_lanes=-1) == ("lane", "lane", "chunked")
    assert fam(deterministic_family=True) == ("lane", "lane", "fused")
    assert fam(storage_dtype=torch.float32) == ("quad", "quad", "chunked")  # fp32-stored messages
    passes = lambda **kw: parity.engine_from_case(g, lib, "cpu", **kw).kernel_family("chunk_passes")  # noqa: E731
    assert passes() == passes(group_lanes=64, backward_mode="chunked") == passes(group_lanes=Q, backward_mode="chunked") == "quad"
    assert passes(group_lanes=-1) == "lane"
    N = parity.pkg._native  # the window's edge, asked of the resolver alone (no buffers)
    p = N.I2cProblem()
    p.abi_version, p.model_id, p.T, p.backward_mode, p.inference, p.dtype = N.ABI_VERSION, N.MODEL_IDS["DoubleCartpoleKnown"], 300, N.BWD_AUTO, N.INF_CUBATURE, 0
    p.group_lanes, p.post_layout, p.gh_degree, p.quad_alpha, p.quad_beta, p.quad_kappa = 0, 0, 3, 1.0, 0.0, 0.0
    import ctypes
    for B, want in ((1, "quad"), (256, "quad"), (257, "lane"), (4096, "lane")):
        p.B = B
        assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_BACKWARD)] == want
        assert lib.i2c_backward_schedule(ctypes.byref(p)) == N.BWD_CHUNKED
    # the compose (+ stitch) pass of that schedule (I2C_SWEEP_CHUNK_PASSES): quad up to 256 trajectories
    for B, want in ((1, "quad"), (256, "quad"), (257, "lane"), (4096, "lane")):
        p.B = B
        assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_PASSES)] == want
    # ... and the stitch pass ALONE keeps the quad form up to 8192 trajectories (I2C_SWEEP_CHUNK_STITCH)
    for B, want in ((768, "quad"), (769, "quad"), (8192, "quad"), (8193, "lane")):
        p.B = B
        assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_STITCH)] == want
    p.B, p.group_lanes = 4096, -1  # one lane per trajectory everywhere: lane passes; a fused walk has no such passes
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_STITCH)] == "lane"
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_PASSES)] == "lane"
    p.B, p.group_lanes = 32768, 0
    assert lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_PASSES) == -2  # I2C_ENOTSUP
    # fp32-stored messages, 4.5 million trajectories, the whole schedule asked for in the quad form: the storage-typed cell windows
    # still fit 2 GiB (the walker is served), the arithmetic-typed composites do not (compose + stitch stay lane kernels)
    p.B, p.group_lanes, p.backward_mode, p.dtype = 4_500_000, 64, N.BWD_CHUNKED, N.F64_F32S
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_BACKWARD)] == "quad"
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_PASSES)] == "lane"
    p.B = 1024
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_CHUNK_PASSES)] == "quad"
    p.group_lanes, p.backward_mode, p.dtype = 0, N.BWD_AUTO, 0
    p.B, p.T = 1, 6  # too short to chunk: the lane kernels' two-pass schedule, as before
    assert N.FAMILY_NAMES[lib.i2c_kernel_family(ctypes.byref(p), N.SWEEP_BACKWARD)] == "lane" and lib.i2c_backward_schedule(ctypes.byref(p)) == N.BWD_TWO_PASS


def test_hostsim_quad_stitch_between_lane_compose_and_lane_walk():
    """The pairing of the BASELINE batches of the d >= 5 models: lane compose pass, QUAD stitch pass, lane walker (one composite
    format, one boundary format). The windows that select it start beyond what the host simulation can run, so a child process
    forces them with the library's experiment knobs (read once per process) and checks two goldens end to end."""
    import os
    import subprocess
    import sys as _sys
    import textwrap

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, "input-inference-for-control_amd")!r}, {os.path.join(ROOT, "tests")!r}]
        import hostsim, parity
        lib = hostsim.load()
        for name, td, ts in (("em_dcp_T60", 1e-7, 1e-6), ("em_quadrotor_T20", 1e-7, 1e-6), ("em_covctrl_qf_T40", 1e-8, 1e-7)):
            eng = parity.check_against_golden(name, lib, "cpu", td, ts, backward_mode="chunked")
            fams = (eng.backward_family, eng.backward_schedule, eng.kernel_family("chunk_passes"), eng.kernel_family("chunk_stitch"))
            assert fams == ("lane", "chunked", "lane", "quad"), fams
        print("ok")
    """)
    env = dict(os.environ, I2C_QUAD_PASSES_MAX_B="0", I2C_QUAD_STITCH_MAX_B="1000")
    r = subprocess.run([_sys.executable, "-c", script], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_hostsim_quad_backward_optional_outputs_and_statistics(lib):
    """The optional outputs of the fused quad walk -- per-cell cost statistics [T][2][B], the smoothed state entering every cell,
    the observation moments -- against the lane walk's on the same forward messages; a ragged batch (5 = one full wave + one
    with three spare slots)."""
    g = load_case("em_dcp_nondiag_T30")
    x0, mu_u = parity.batched_inputs(g, 5)
    out = {}
    for key, kw in (("quad", dict(group_lanes=64)), ("lane", dict(group_lanes=64, backward_mode="fused"))):
        eng = parity.engine_from_case(g, lib, "cpu", x0=x0, mu_u=mu_u, keep_xm=True, keep_zpost=True, **kw)
        if key == "lane":  # (a quad request with the fused walk asked for IS the quad walk: pin the lane one through the lane family)
            eng = parity.engine_from_case(g, lib, "cpu", x0=x0, mu_u=mu_u, keep_xm=True, keep_zpost=True, group_lanes=-1, backward_mode="fused")
        eng.cell_stats = torch.zeros(eng.H, 2, eng.B, dtype=torch.float64)
        eng.forward_backward()
        assert eng.backward_family == key and eng.failures() == []
        out[key] = [parity.np_(x) for x in (eng.cell_stats, eng.term_stats, *eng.smoothed_next_state(), *eng.observed_marginal(), *eng.terminal_observed_marginal())]
        np.testing.assert_allclose(out[key][0].sum(0), out[key][1][1:3], rtol=1e-12)
    for a, b in zip(out["quad"], out["lane"]):
        assert_close(a, b, 1e-7, "quad backward walk vs lane backward walk")


@pytest.mark.parametrize("name,B", [("em_dcp_T60", 5), ("em_cartpole_T100", 3), ("em_pendulum_T200", 6)])
def test_hostsim_quad_forward_general_weights_batch_vs_oracle(lib, name, B):
    """The GENERAL variant (cubature weights with lam != 0, W != 1) on the d = 5 / 7 models against the batched oracle with the same
    rule -- and the default family of such a problem inside the quad window is the quad kernel now (it was the group fallback)."""
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", B, 2, tol=1e-7, quad=(1.2, 0.44, 0.5))
    assert eng.forward_family == ("quad" if name != "em_pendulum_T200" else "lane")
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", B, 2, tol=1e-7, quad=(1.0, 0.0, 0.5), group_lanes=64)  # (W = 1, but a weight on the centre: lam = 0.5)
    assert eng.forward_family == "quad" and eng.backward_family == "quad"  # (round 6: the GENERAL moments in the backward walk too)


@pytest.mark.parametrize("name,B", [("em_dcp_T60", 6), ("em_pendulum_T200", 9), ("em_quadrotor_T20", 5), ("em_cartpole_T100", 3)])
def test_hostsim_quad_forward_batch_vs_oracle(lib, name, B):
    """Ragged batches (not a multiple of the four trajectories of a wavefront: the spare slots repeat the last trajectory and
    store nothing) against the batched oracle."""
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", B, 2, tol=1e-7, group_lanes=64)
    assert eng.forward_family == "quad" and eng.backward_family == "quad"


@pytest.mark.parametrize("name", QUAD12)
def test_hostsim_quad12_quad_forward_vs_reference_golden(lib, name):
    """The quad kernels on the 12-state quadrotor (d = 16: sixteen pair rows, two evaluation passes, four pivot blocks per
    factorisation; the backward sweep as the fused walk of four trajectories per wavefront), asked for with LANES_QUAD next to
    the model's wave kernels; forward messages trajectory-major. The last case: covariance control (tempered terminal prior)."""
    eng = parity.check_against_golden(name, lib, "cpu", 1e-7, 1e-6, group_lanes=parity.pkg._native.LANES_QUAD)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad") and eng.fwd_trajectory_major


@pytest.mark.parametrize("name", QUAD12)
def test_hostsim_quad12_quad_forward_wave_backward_vs_reference_golden(lib, name):
    """An explicit two-pass request keeps the wave backward sweep behind the quad forward sweep (same buffers)."""
    eng = parity.check_against_golden(name, lib, "cpu", 1e-7, 1e-6, group_lanes=parity.pkg._native.LANES_QUAD, backward_mode="two_pass")
    assert (eng.forward_family, eng.backward_family) == ("quad", "wave") and eng.fwd_trajectory_major


def test_hostsim_quad12_quad_backward_cell_statistics(lib):
    """The per-cell cost statistics (an optional output of the fused walks: [T][2][B]) of the quad backward sweep add up to the
    sums it reports, and agree with the wave form's."""
    stats = {}
    for lanes in (parity.pkg._native.LANES_QUAD, 64):
        eng = parity.engine_from_case(load_case("em_quad12_nondiag_T12"), lib, "cpu", group_lanes=lanes)
        eng.cell_stats = torch.zeros(eng.H, 2, eng.B, dtype=torch.float64)
        eng.forward_backward()
        stats[lanes] = (parity.np_(eng.cell_stats), parity.np_(eng.term_stats))
        np.testing.assert_allclose(stats[lanes][0].sum(0)[:, 0], stats[lanes][1][1:3, 0], rtol=1e-12)
    np.testing.assert_allclose(stats[64][0], stats[parity.pkg._native.LANES_QUAD][0], rtol=1e-9)


@pytest.mark.parametrize("name", ["em_quad12_covctrl_T12", "em_quad12_nondiag_T12"])
def test_hostsim_quad12_quad_sweeps_batch_vs_oracle(lib, name):
    """Covariance control and non-diagonal cost weights on a ragged batch (five trajectories: a full wavefront slot set and one
    with three spare slots), both sweeps on the quad kernels."""
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", 5, 2, tol=1e-7, group_lanes=parity.pkg._native.LANES_QUAD)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad")


def test_hostsim_quad12_quad_forward_batch_vs_oracle(lib):
    eng, _ = parity.check_batch_against_oracle("em_quad12_T20", lib, "cpu", 5, 2, tol=1e-7, group_lanes=parity.pkg._native.LANES_QUAD)
    assert eng.forward_family == "quad"


@pytest.mark.parametrize("name,lanes,family", [("em_quad12_T20", 64, "wave"), ("em_quad12_T20", 164, "quad"), ("em_quadrotor_T20", 64, "quad")])
@pytest.mark.parametrize("scale", [1e-3, 1e3])
def test_hostsim_square_root_update_at_extreme_temperatures(lib, name, lanes, family, scale):
    """The identity-observation models update the FACTOR of the joint prior (w_kalman_sqrt / q_kalman_sqrt: chol(I + L^T N^-1 L) from
    the last row upwards) instead of factoring sig_0 + N and the updated joint. N = alpha * xi: a temperature 10^3 times smaller
    makes N^-1 dominate that matrix, one 10^3 times larger leaves it next to the identity -- both against the oracle, which solves
    the reference's covariance form (i2c.py:394-403). (At 10^-4 the last cell's gain, ~1e-7 of the largest, differs from the oracle's by
    8e-4 of itself on EVERY kernel family, the classical group form included: the comparison's floor, not this update.)"""
    import json

    from golden_util import load_case

    alpha = load_case(name).meta["alpha"] * scale
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", 5, 2, tol=1e-6, group_lanes=lanes, meta_override={"alpha": alpha})
    assert eng.forward_family == family


def test_quad_forward_refuses_what_it_does_not_cover(lib):
    """General cubature weights (a weight on the centre point, weights that need not sum to one) are in the quad form for every
    model of the d <= 8 geometry (round 5: where every observation goes through the sigma points; round 6: the identity-observation
    models too) and for d = 16 (round 6); the wave kernels and the other inference rules are refused with I2C_ENOTSUP when asked for."""
    with pytest.raises(RuntimeError, match="-2"):  # the wave kernels (64 on the d = 16 model) only have the unit rule
        parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu", quad=(1.2, 0.44, 0.5), group_lanes=64)
    with pytest.raises(RuntimeError, match="-2"):  # Linearize() on the quad kernels
        parity.engine_from_case(load_case("lin_pendulum_T100"), lib, "cpu", group_lanes=64)
    with pytest.raises(ValueError):  # d = 16 has the wave kernels under 64; models without either refuse it
        parity.engine_from_case(load_case("em_pendulum_T200"), lib, "cpu", group_lanes=32)


@pytest.mark.parametrize("name,tol_d,tol_s", LANE_GOLDEN)
def test_hostsim_lane_kernels_vs_reference_golden(lib, name, tol_d, tol_s):
    parity.check_against_golden(name, lib, "cpu", tol_d, tol_s, group_lanes=-1)


def test_hostsim_group_kernels_batch_vs_oracle(lib):
    parity.check_batch_against_oracle("em_quad12_T20", lib, "cpu", 5, 3, tol=1e-7)  # wave kernels (the default)
    parity.check_batch_against_oracle("em_quad12_T20", lib, "cpu", 5, 3, tol=1e-7, group_lanes=16)
    parity.check_batch_against_oracle("em_dcp_T60", lib, "cpu", 3, 3, tol=1e-7, group_lanes=True)


def test_group_kernels_refuse_what_they_do_not_cover(lib):
    """The other inference rules are not available in the group form: the library says I2C_ENOTSUP instead of computing
    something else. (Non-diagonal cost weights were refused until round 3: now covered, goldens em_*_nondiag_*.)"""
    from golden_util import load_case

    eng = parity.engine_from_case(load_case("lin_pendulum_T100"), lib, "cpu", keep_xm=True)
    eng._problem.group_lanes = 4  # behind the engine's own argument check
    with pytest.raises(RuntimeError, match="-2"):
        eng.forward_sweep()
    with pytest.raises(ValueError):
        parity.engine_from_case(load_case("em_pendulum_T200"), lib, "cpu", group_lanes=16)
    with pytest.raises(ValueError):  # d = 16 has no one-lane kernels
        parity.engine_from_case(load_case("em_quad12_T20"), lib, "cpu", group_lanes=-1)


def test_group_kernels_dense_weights_match_lane_kernels(lib):
    """A dense Q on the pendulum: the group form's general-weight cost (g_cost_full) against the lane kernels' gaussian_cost."""
    import numpy as np
    import torch
    from golden_util import Case, load_case

    g = load_case("em_pendulum_T40_quad_general")
    dense = Case({**g, "Q": np.array([[2.0, 0.3, -0.1], [0.3, 50.0, 0.4], [-0.1, 0.4, 1.5]]),
                  "Qf": np.array([[1.0, 0.2, 0.0], [0.2, 20.0, -0.3], [0.0, -0.3, 2.5]])})
    runs = []
    for gl in (0, True):
        eng = parity.engine_from_case(dense, lib, "cpu", group_lanes=gl)
        for _ in range(3):
            eng.learn_msgs()
        assert eng.failures() == []
        runs.append(eng)
    a, b = runs
    assert (a.forward_family, b.forward_family) == ("lane", "group")
    for x, y in zip(a.costs_m + a.costs_m_var + a.alphas, b.costs_m + b.costs_m_var + b.alphas):
        assert torch.allclose(x, y, rtol=1e-10, atol=0)
    assert torch.allclose(a.post, b.post, rtol=1e-9, atol=1e-12)


LINEARIZE = [
    ("lin_linear_T60", 1e-8, 1e-7),
    ("lin_covctrl_T50", 1e-8, 1e-7),
    ("lin_covctrl_qf_T30", 1e-8, 1e-7),  # + a terminal cost: the back-calculated sig_xi_terminal (i2c.py:455-462)
    ("lin_pendulum_T100", 1e-8, 1e-6),
    ("lin_pendulum_T40_propagate", 1e-8, 1e-6),  # + closed-loop propagation (unit cubature rule), expert controller
    ("lin_cartpole_T100", 1e-7, 1e-6),
    ("lin_dcp_T80", 1e-7, 1e-6),
    ("lin_quad12_T20", 1e-7, 1e-6),  # d = 16: the wave kernels' Linearize variant (dual-number Jacobian, one input per lane)
]


@pytest.mark.parametrize("name,tol_d,tol_s", LINEARIZE)
def test_hostsim_linearize_vs_reference_golden(lib, name, tol_d, tol_s):
    """Linearize() inference (i2c.py:244-348, 449-542): Jacobians by forward-mode differentiation of the model
    functors, against the reference's captured runs (linear systems: exact pin; nonlinear: see the oracle header)."""
    parity.check_against_golden(name, lib, "cpu", tol_d, tol_s)


@pytest.mark.parametrize("name", ["gh3_pendulum_T40", "gh4_linear_T30", "gh3_covctrl_T100"])
def test_hostsim_gauss_hermite_vs_reference_golden(lib, name):
    """GaussHermiteQuadrature(degree) inference (exp_types.py:52-68): tensor-grid transform in the forward, backward
    and propagation kernels, against the reference's captured runs."""
    parity.check_against_golden(name, lib, "cpu", 1e-8, 1e-7)


def test_hostsim_gauss_hermite_batch_vs_oracle(lib):
    parity.check_batch_against_oracle("gh3_pendulum_T40", lib, "cpu", 5, 3, tol=1e-7)


@pytest.mark.parametrize("name,B,iters", [("lin_pendulum_T100", 6, 4), ("lin_dcp_T80", 3, 3), ("lin_quad12_T20", 3, 3)])
def test_hostsim_linearize_batch_vs_oracle(lib, name, B, iters):
    parity.check_batch_against_oracle(name, lib, "cpu", B, iters, tol=1e-7)


@pytest.mark.parametrize("name,B,iters", [("em_pendulum_T200", 8, 4), ("em_dcp_T60", 4, 3), ("em_covctrl_T100", 4, 4)])
def test_hostsim_batch_vs_oracle(lib, name, B, iters):
    parity.check_batch_against_oracle(name, lib, "cpu", B, iters, tol=1e-7)


@pytest.mark.parametrize("mode", ["fused", "two_pass", "chunked"])
@pytest.mark.parametrize("name", ["em_pendulum_T200", "em_covctrl_T100", "em_dcp_T60"])
def test_hostsim_backward_schedules_match_reference(lib, name, mode):
    """Every schedule of the backward sweep (fused single pass / scan + per-cell / chunk-composed) against
    the same golden vectors; the default `auto` picks chunked here and is what the other tests run."""
    parity.check_against_golden(name, lib, "cpu", 1e-7, 1e-6, n_iters=4, backward_mode=mode)


def test_hostsim_general_weights_sum_not_one(lib):
    """CubatureQuadrature with 1 - alpha^2 + beta != 0 (weights do not sum to 1): the correction
    terms of the centred accumulation against the oracle's literal reference formula."""
    import numpy as np
    from golden_util import Case, load_case

    g = load_case("em_pendulum_T40_quad_general")
    meta = dict(g.meta, quad=[1.0, -0.03, 0.4], T=12)
    g2 = Case({**g, "meta": np.array(__import__("json").dumps(meta)), "mu_u": g["mu_u"][:12]})
    x0, mu_u = parity.batched_inputs(g2, 3)
    eng = parity.engine_from_case(g2, lib, "cpu", x0=x0, mu_u=mu_u)
    from golden_util import oracle_from_case

    o = oracle_from_case(Case({**g2, "mu_u": mu_u}), x0=x0)
    for it in range(3):
        eng.learn_msgs()
        o.learn_msgs()
        mu, sig = eng.marginal_state_action()
        parity.assert_close(parity.np_(mu), o.mu_xu0_m, 1e-9, f"W!=1 it{it} mu")
        parity.assert_close(parity.np_(sig), o.sig_xu0_m, 1e-9, f"W!=1 it{it} sig")
        parity.assert_close(parity.np_(eng.alpha), o.alpha, 1e-9, f"W!=1 it{it} alpha")


def test_hostsim_fused_em_loop_matches_stepwise(lib):
    """i2c_learn (N iterations enqueued from C++) == N x learn_msgs(), including the FF -> FB switch."""
    import torch
    from golden_util import load_case

    g = load_case("em_pendulum_T40_quad_general")
    x0, mu_u = parity.batched_inputs(g, 5)
    a = parity.engine_from_case(g, lib, "cpu", x0=x0, mu_u=mu_u)
    b = parity.pkg.BatchedI2c(parity.product_model(g), g.meta["T"], g["Q"], g["R"], g["Qf"], g.meta["alpha"], g.meta["tol"],
                              mu_u, g["sig_u"], quad=tuple(g.meta["quad"]), x0=x0, lib=lib, device="cpu")
    for _ in range(5):
        a.learn_msgs()
    b.learn(5)
    assert torch.equal(a.post, b.post) and torch.equal(a.alpha, b.alpha) and torch.equal(a.feedforward, b.feedforward)
    assert b.em_iter == 5 and len(b.alphas) == 6 and len(b.costs_m) == 5
    assert torch.equal(torch.stack(a.costs_m), torch.stack(b.costs_m))
    assert torch.equal(torch.stack(a.alphas), torch.stack(b.alphas))


def _post_layout_is_transparent(lib, device, monkeypatch):
    """The posterior / prior buffers of the wave-capable models are stored trajectory-major (I2cProblem.post_layout = 1); the
    standard [T][e][B] layout stays available (post_layout=0). Same arithmetic, different addressing: EM iterations on both
    kernel families, closed-loop propagation, the MPC step with its ring shift, and policy rollouts must agree bit for bit."""
    import numpy as np
    import torch
    from golden_util import load_case

    g = load_case("em_quad12_T12_propagate")
    x0, mu_u = parity.batched_inputs(g, 5)
    runs = {}
    for layout in ("1", "0"):
        for lanes in (0, 16):
            e = parity.engine_from_case(g, lib, device, x0=x0, mu_u=mu_u, group_lanes=lanes, post_layout=int(layout))
            assert e.post_layout == int(layout) and e.post.shape == (g.meta["T"], e.dims.e_post, 5)
            assert e.post.stride()[1] == (1 if layout == "1" else 5)
            e.propagate()
            for _ in range(3):
                e.learn_msgs()
            e.enable_per_cell_alpha()
            sig_zeta = 1e-4 * np.eye(9)
            y = torch.as_tensor(np.ascontiguousarray(parity.product_model(g).measure(x0).T), dtype=torch.float64, device=device)
            u = torch.as_tensor(np.ascontiguousarray(mu_u[:, 0, :].T), dtype=torch.float64, device=device)
            act = [e.mpc_step(2, y, u, sig_zeta)[0].clone() for _ in range(2)]
            roll = e.rollout(2, "expert_soft", process_noise=False)
            assert e.failures() == []
            runs[(layout, lanes)] = (e.post.clone(), e.prop.clone(), torch.stack(act), roll["xu"].clone(), e.alpha.clone())
    for lanes in (0, 16):
        for k, (a, b) in enumerate(zip(runs[("1", lanes)], runs[("0", lanes)])):
            if k == 1 and lanes == 0:
                # the automatic pick propagates d=16 with the quad kernel on the trajectory-major layout and with the group
                # kernel on the standard one: the same recursion in a different summation order
                assert torch.allclose(a, b, rtol=1e-9, atol=1e-12)
            else:
                assert torch.equal(a, b)


def test_hostsim_post_layout_is_transparent(lib, monkeypatch):
    _post_layout_is_transparent(lib, "cpu", monkeypatch)


def _terminal_prior_with_terminal_cost(lib, device, name, T, mu_T, sig_T, family, tol):
    """Covariance control (a terminal state prior, i2c.py:548-559 / 453-472) TOGETHER with a terminal cost, against the batched
    oracle. Under the cubature rule the tempered product at the end of the chain (wave kernels: its Kalman form); under Linearize() the smoothed terminal state is pinned to the prior and sig_z3_m carries the back-calculated
    sig_xi_terminal (i2c.py:455-462, 499-501), which reaches alpha through the terminal statistic (:989-992)."""
    import json

    import numpy as np
    from golden_util import Case, assert_close, load_case, oracle_from_case

    g = load_case(name)
    assert "Qf" in g
    meta = dict(g.meta, T=T)
    case = Case({**g, "meta": np.array(json.dumps(meta)), "mu_u": g["mu_u"][:T], "mu_x_term": np.asarray(mu_T, float),
                 "sig_x_term": np.asarray(sig_T, float)})
    x0, mu_u = parity.batched_inputs(case, 3)
    eng = parity.engine_from_case(case, lib, device, x0=x0, mu_u=mu_u)
    assert eng.forward_family == family
    o = oracle_from_case(Case({**case, "mu_u": mu_u}), x0=x0)
    for it in range(2):  # (the back-calculation amplifies rounding-level differences of sig_x3_f by cond(S3f - S_T)^2: two sweeps)
        eng.learn_msgs()
        o.learn_msgs()
        if meta.get("inference") == "linearize" and eng.nzt == eng.nx:
            # the kernel's multiplier from the kernel's OWN filtered terminal covariance (identity observation, i2c.py:455-462)
            S3f = parity.np_(eng.forward_messages()["sig_x3_f"])[:, T - 1]
            dS = S3f - case["sig_x_term"]
            if it == 1:  # neither tighter nor looser than the filtered covariance: the multiplier is indefinite
                assert np.linalg.eigvalsh(dS[0]).min() < 0 < np.linalg.eigvalsh(dS[0]).max()
            want = case["sig_x_term"] + S3f @ np.linalg.inv(dS) @ S3f - S3f
            assert_close(parity.np_(eng.terminal_observed_marginal()[1]), want, 1e-9 * np.linalg.cond(dS).max() ** 2,
                         f"{name} terminal prior it{it}: sig_z3_m from the kernel's own sig_x3_f")
        mu, sig = eng.marginal_state_action()
        assert_close(parity.np_(mu), o.mu_xu0_m, tol, f"{name} terminal prior it{it}: mu")
        assert_close(parity.np_(sig), o.sig_xu0_m, tol, f"{name} terminal prior it{it}: sig")
        mzt, szt = eng.terminal_observed_marginal()
        assert_close(parity.np_(mzt), o.mu_z3_m, tol, f"{name} terminal prior it{it}: mu_z3_m")
        assert_close(parity.np_(szt), o.sig_z3_m, tol * 10, f"{name} terminal prior it{it}: sig_z3_m")
        assert_close(parity.np_(eng.alpha), o.alpha, tol * 10, f"{name} terminal prior it{it}: alpha")
    assert eng.failures() == []


Q12_TERM = ([0.3, -0.2, 0.5] + [0.0] * 9, np.diag([1e-2] * 3 + [1e-1] * 9))
TERMINAL_PRIOR = [
    ("em_quad12_T20", 8, *Q12_TERM, "wave"),  # (round 4: the wave backward sweep has the tempered terminal prior; group kernels until then)
    ("lin_quad12_T20", 8, *Q12_TERM, "wave"),
    ("lin_linear_T60", 30, [1.0, 0.5], [[1e-3, 2e-4], [2e-4, 50.0]], "lane"),
]


@pytest.mark.parametrize("name,T,mu_T,sig_T,family", TERMINAL_PRIOR)
def test_hostsim_terminal_prior_with_terminal_cost_vs_oracle(lib, name, T, mu_T, sig_T, family):
    _terminal_prior_with_terminal_cost(lib, "cpu", name, T, mu_T, sig_T, family, 1e-7)


def _linearize_chunked_equals_sequential(lib, device, name, rtol):
    """The chunked form of the Linearize backward sweep (the default of small batches) against the sequential walk."""
    g = load_case(name)
    runs = []
    for mode in ("fused", "chunked"):
        eng = parity.engine_from_case(g, lib, device, backward_mode=mode)
        assert eng.backward_schedule == mode
        for _ in range(4):
            eng.learn_msgs()
        assert eng.failures() == []
        runs.append(eng)
    a, b = runs
    from golden_util import rel_err  # (max-norm relative: the composites are applied in another order, rounding differs)

    for x, y in zip(a.costs_m + a.costs_m_var + a.alphas, b.costs_m + b.costs_m_var + b.alphas):
        assert rel_err(parity.np_(x), parity.np_(y)) <= rtol
    mu_a, sig_a = a.marginal_state_action()
    mu_b, sig_b = b.marginal_state_action()
    assert rel_err(parity.np_(mu_a), parity.np_(mu_b)) <= rtol and rel_err(parity.np_(sig_a), parity.np_(sig_b)) <= rtol
    for x, y in zip(a.local_linear_policy(), b.local_linear_policy()):
        assert rel_err(parity.np_(x), parity.np_(y)) <= rtol * 10
    assert rel_err(parity.np_(a.term_stats), parity.np_(b.term_stats)) <= rtol
    assert parity.engine_from_case(g, lib, device).backward_schedule == "chunked"  # ... and it is the default here


@pytest.mark.parametrize("name", ["lin_pendulum_T100", "lin_cartpole_T100", "lin_dcp_T80", "lin_covctrl_qf_T30"])
def test_hostsim_linearize_chunked_backward_equals_sequential(lib, name):
    _linearize_chunked_equals_sequential(lib, "cpu", name, 1e-9)


@pytest.mark.parametrize("mode", ["auto", "chunked"])
def test_hostsim_quad_backward_failure_is_per_trajectory(lib, mode):
    """A smoothed joint that is not positive definite in ONE cell of ONE trajectory (its filtered variance poisoned between the
    sweeps) is reported as that trajectory's failure -- reason 7 (the backward cell), that cell -- and leaves the other three
    trajectories of its wavefront, and the rest of the batch, untouched. The fused quad walk, and the chunked schedule with its
    compose, stitch and walk passes in the quad form (the poisoned cell then also enters a composite: its chunk and the ones the
    stitch pass reaches after it belong to that trajectory alone)."""
    g = load_case("em_dcp_T60")
    x0, mu_u = parity.batched_inputs(g, 6)
    eng = parity.engine_from_case(g, lib, "cpu", x0=x0, mu_u=mu_u, group_lanes=64, backward_mode=mode)
    clean = parity.engine_from_case(g, lib, "cpu", x0=x0, mu_u=mu_u, group_lanes=64, backward_mode=mode)
    for e in (eng, clean):
        e.forward_sweep()
    eng.fwd[17, eng.d, 2] = -1.0  # sig_xu1_f[0][0] of cell 17, trajectory 2
    for e in (eng, clean):
        e.backward_sweep()
    assert eng.backward_family == "quad" and clean.failures() == []
    if mode == "auto":
        assert eng.failures() == [(2, 7, 17)]
    else:  # (the chunks walk concurrently: whichever cell of trajectory 2 noticed first is recorded)
        assert eng.kernel_family("chunk_passes") == "quad" and [(b, r) for b, r, _ in eng.failures()] == [(2, 7)]
    ok = [0, 1, 3, 4, 5]
    for a, b in zip(eng.marginal_state_action() + eng.local_linear_policy(), clean.marginal_state_action() + clean.local_linear_policy()):
        assert torch.equal(a[ok], b[ok])
    assert not torch.isfinite(eng.local_linear_policy()[0][2, 17]).all()  # the failed cell's controller is NaN, not silently wrong


def test_hostsim_quad_sweeps_minimum_energy_model_vs_oracle(lib):
    """LinearKnownMinimumEnergy (only the action is observed, identity terminal observation, covariance control with a tempered
    terminal prior; env_def.py:173-230) ships with a Linearize golden only: the same problem under the cubature rule, both sweeps on
    the quad kernels, against the batched oracle."""
    eng, _ = parity.check_batch_against_oracle("lin_covctrl_T50", lib, "cpu", 6, 3, tol=1e-7 if "cpu" == "cpu" else 1e-6, group_lanes=64,
                                               meta_override={"inference": "cubature", "quad": [1.0, 0.0, 0.0]})
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad")


@pytest.mark.parametrize("name,B", [("em_pendulum_T200", 3), ("em_cartpole_T100", 3), ("em_dcp_T60", 3)])
def test_hostsim_quad_sweeps_weights_that_do_not_sum_to_one(lib, name, B):
    """CubatureQuadrature(1.05, 0, 0.3): W = sum of weights_sig = 2 - alpha^2 + beta = 0.8975 -- the reference weighs the mean, the
    covariance AND the process noise (sum_p w_p sig_eta = W sig_eta, quadrature.py:57) with it. Round 6 found the quad forward kernel's
    GENERAL variant adding the UNWEIGHTED sig_eta (every earlier general-weights case -- (1.2, 0.44, 0.5), (1, 0, 0.5) -- happens to have
    W = 1): up to 0.35 relative on the double cartpole. Both quad sweeps against the batched oracle with the same rule. (W > 1 is not a
    usable regime: the reference's own covariances lose positive definiteness.)"""
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", B, 2, tol=1e-7, quad=(1.05, 0.0, 0.3), group_lanes=64)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad")


@pytest.mark.parametrize("name,quad", [("em_quadrotor_T20", (1.2, 0.44, 0.5)), ("em_linear_T60", (1.2, 0.44, 0.5)), ("em_linear_T60", (1.05, 0.0, 0.3)),
                                       ("lin_covctrl_T50", (1.2, 0.44, 0.5))])
def test_hostsim_quad_sweeps_general_weights_identity_observation_models(lib, name, quad):
    """Round 6 (review item 8, the d <= 8 part): CubatureQuadrature(alpha, beta, kappa) with lam != 0 / W != 1 on the models whose
    observation is the joint itself (planar quadrotor: d = 8, no spare pair row -- the centre is an extra evaluation pass; linear models;
    the minimum-energy model's identity TERMINAL observation): exact moments W m, S + (W - W^2) m m^T through the general update in the
    forward sweep, the same moments in the backward walk's cost -- both quad sweeps against the batched oracle."""
    kw = dict(meta_override={"inference": "cubature"}) if name.startswith("lin_") else {}
    eng, _ = parity.check_batch_against_oracle(name, lib, "cpu", 5, 2, tol=1e-6, quad=quad, group_lanes=64, **kw)
    assert (eng.forward_family, eng.backward_family) == ("quad", "quad")
    if name == "em_quadrotor_T20":  # ... and it is the DEFAULT forward sweep of such a problem inside the quad window (it was the group fallback)
        assert parity.engine_from_case(load_case(name), lib, "cpu", quad=quad).forward_family == "quad"


@pytest.mark.parametrize("name", ["em_quad12_T20", "em_quad12_T12_propagate", "em_quad12_covctrl_T12", "em_quad12_nondiag_T12"])
@pytest.mark.parametrize("quad", [(1.2, 0.44, 0.5), (1.05, 0.0, 0.3)])
def test_hostsim_quad12_general_weights_run_on_the_quad_kernels(lib, name, quad):
    """Round 6 (review item 8, the d = 16 part): any CubatureQuadrature(alpha, beta, kappa) on the 12-state quadrotor is the quad
    kernels' at EVERY batch size -- forward sweep (general identity update, the centre point as a third evaluation pass), backward
    walk and closed-loop propagation (the identity observation's exact moments W m, S + (W - W^2) m m^T in the expected cost) --
    where it used to fall back to the group kernels, whose waves serialise beyond 1024. Against the batched oracle with the same
    rule, propagated quantities included."""
    eng, o = parity.check_batch_against_oracle(name, lib, "cpu", 5, 2, tol=1e-6, quad=quad)
    assert (eng.forward_family, eng.backward_family, eng.kernel_family("propagate")) == ("quad", "quad", "quad")
    if eng._propagate:
        p = eng.propagated()
        for key in ("mu_xu0_pf", "sig_xu0_pf", "mu_x3_pf", "sig_x3_pf"):
            parity.close(parity.np_(p[key]), getattr(o, key), 1e-6, f"{name} {key}")
        parity.close(parity.np_(eng.costs_pf[-1]), o.costs_pf[-1], 1e-6, f"{name} propagated cost")
