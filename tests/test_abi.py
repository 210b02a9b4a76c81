"""The C-ABI shared library loads (no GPU needed) and exports every function that
include/i2c_hip.h declares; i2c_query reports the compile-time layout of every model."""
import ctypes
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("input-inference-for-control_amd")


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "i2c_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(i2c_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(pkg._native.EXPORTED_SYMBOLS)


def test_hip_library_exports_the_whole_abi():
    lib = pkg.load_library()  # in-tree gfx950 build (built by __graft_entry__.build())
    assert os.path.basename(lib.path) == "libi2c_hip.so" and not lib.is_host_sim
    raw = ctypes.CDLL(lib.path)
    for name in _declared_functions():
        assert hasattr(raw, name), name
    assert lib.i2c_abi_version() == pkg._native.ABI_VERSION
    assert "gfx950" in lib.build_info


def test_query_layouts():
    lib = pkg.load_library()
    sym = lambda n: n * (n + 1) // 2  # noqa: E731
    expect = {0: (2, 1, 4, 3), 1: (2, 1, 1, 0), 2: (4, 1, 6, 5), 3: (6, 1, 9, 8), 4: (2, 1, 3, 2), 5: (2, 1, 1, 2),
              6: (6, 2, 8, 6)}
    for mid, (nx, nu, nz, nzt) in expect.items():
        d = lib.query(mid)
        assert (d.nx, d.nu, d.nz, d.nzt) == (nx, nu, nz, nzt)
        dd = nx + nu
        assert d.e_post == dd + sym(dd) + nu * nx + nu + sym(nu)
        assert d.e_fwd == dd + sym(dd) + nx + sym(nx) + dd * nx
        assert d.e_xm == nx + sym(nx) and d.e_zpost == nz + sym(nz)
    # 64 elements (512 B fp64) per pendulum cell-iteration, 292 for the double cartpole
    for mid, total in ((0, 64), (3, 292)):
        d = lib.query(mid)
        pri = d.e_post - d.nu - sym(d.nu)
        assert pri + 2 * d.e_fwd + d.e_post == total
    import pytest

    with pytest.raises(ValueError):
        lib.query(99)


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    lib = pkg.load_library()
    p = pkg._native.I2cProblem()
    assert lib.i2c_forward_sweep(ctypes.byref(p), None, None, None, None, None) == -1  # I2C_EINVAL
    assert lib.i2c_mstep(ctypes.byref(p), None, 0.0, 1, None, None) == -1


def _shape(model_id, B, T, mode=0, inference=0, dtype=0, group_lanes=0, post_layout=0, gh_degree=3):
    """An I2cProblem with its scalar fields only: what the two resolvers read (no buffer exists yet)."""
    N = pkg._native
    p = N.I2cProblem()
    p.abi_version, p.model_id, p.B, p.T, p.backward_mode, p.inference, p.dtype = N.ABI_VERSION, model_id, B, T, mode, inference, dtype
    p.group_lanes, p.post_layout, p.gh_degree = group_lanes, post_layout, gh_degree
    p.quad_alpha, p.quad_beta, p.quad_kappa = 1.0, 0.0, 0.0
    return p


def test_backward_schedule_rule():
    """i2c_backward_schedule(problem) is THE resolver of the backward schedule: the lane kernels' batch rule (I2C_BWD_AUTO: chunked
    below the model's crossover batch (12288; double cartpole 20480), fused from there on; chunked needs T >= 8; explicit requests honoured), and on top of it the family
    of the sweep (wave / quad / group: the fused walk), the inference rule (Linearize, Gauss-Hermite: fused, chunked at small
    batches) and the storage type; refusals come back as the error code the sweep would return."""
    lib = pkg.load_library()
    N = pkg._native

    def f(*a, **k):
        return lib.i2c_backward_schedule(ctypes.byref(_shape(*a, **k)))

    assert f(0, 4096, 200, N.BWD_AUTO) == N.BWD_CHUNKED
    assert f(0, 32768, 200, N.BWD_AUTO) == N.BWD_FUSED and f(0, 8192, 200, N.BWD_AUTO) == N.BWD_CHUNKED
    assert f(2, 65536, 500, N.BWD_AUTO) == N.BWD_FUSED       # cartpole, d = 5
    assert f(3, 32768, 300, N.BWD_AUTO) == N.BWD_FUSED       # double cartpole, d = 7 (its fused walk no longer spills)
    assert f(6, 32768, 50, N.BWD_AUTO) == N.BWD_FUSED        # quadrotor, d = 8: fits since the rows are single-buffered
    assert f(3, 4096, 300, N.BWD_AUTO) == N.BWD_CHUNKED
    # the crossover is per model, re-derived from time and HBM traffic in round 5 (profiles/r5_backward_crossover.txt)
    assert f(0, 12287, 200) == N.BWD_CHUNKED and f(0, 12288, 200) == N.BWD_FUSED      # pendulum-sized models
    assert f(3, 20479, 300) == N.BWD_CHUNKED and f(3, 20480, 300) == N.BWD_FUSED      # double cartpole
    assert f(2, 12287, 500) == N.BWD_CHUNKED and f(2, 12288, 500) == N.BWD_FUSED      # cartpole: the library-wide default
    assert f(0, 4096, 5, N.BWD_AUTO) == N.BWD_TWO_PASS
    assert f(0, 4096, 5, N.BWD_CHUNKED) == N.BWD_TWO_PASS
    for m in (N.BWD_TWO_PASS, N.BWD_FUSED, N.BWD_CHUNKED):
        assert f(3, 100000, 300, m) == m
    assert f(99, 4096, 200, N.BWD_AUTO) == -1 and f(0, 4096, 200, 7) == -1 and f(0, 0, 200, 0) == -1  # I2C_EINVAL
    assert lib.i2c_backward_schedule(None) == -1
    # fp32-stored messages (I2C_F64_F32S): the lane rule
    assert f(0, 4096, 200, dtype=N.F64_F32S) == N.BWD_CHUNKED and f(0, 65536, 200, dtype=N.F64_F32S) == N.BWD_FUSED
    # Linearize(): chunked at small batches (fp64 storage), the sequential walk otherwise; no two-pass form
    assert f(0, 4096, 200, inference=N.INF_LINEARIZE) == N.BWD_CHUNKED
    assert f(0, 65536, 200, inference=N.INF_LINEARIZE) == N.BWD_FUSED
    assert f(0, 4096, 200, N.BWD_TWO_PASS, inference=N.INF_LINEARIZE) == N.BWD_FUSED
    assert f(1, 4096, 100, inference=N.INF_LINEARIZE) == -1   # PendulumKnownActReg has no terminal observation (i2c.py:500-501)
    assert f(0, 4096, 200, inference=N.INF_LINEARIZE, dtype=N.F64_F32S) == -2  # I2C_ENOTSUP
    # Gauss-Hermite: the same pair of schedules
    assert f(0, 4096, 200, inference=N.INF_GAUSS_HERMITE) == N.BWD_CHUNKED
    assert f(0, 65536, 200, inference=N.INF_GAUSS_HERMITE) == N.BWD_FUSED
    assert f(0, 4096, 200, inference=N.INF_GAUSS_HERMITE, gh_degree=0) == -1
    # group kernels: one schedule
    assert f(3, 4096, 300, group_lanes=16) == N.BWD_FUSED and f(0, 4096, 200, N.BWD_CHUNKED, group_lanes=4) == N.BWD_FUSED
    assert f(0, 4096, 200, group_lanes=8) == -2               # not this model's group width
    # the 12-state quadrotor (trajectory-major posterior): wave kernels -> fused, two-pass on request; quad forward above 1024, quad backward above 2048 trajectories;
    # Linearize keeps the wave form; group kernels when asked for
    Q12 = 7
    assert f(Q12, 1024, 50, post_layout=1) == N.BWD_FUSED and f(Q12, 1024, 50, N.BWD_CHUNKED, post_layout=1) == N.BWD_FUSED
    assert f(Q12, 1024, 50, N.BWD_TWO_PASS, post_layout=1) == N.BWD_TWO_PASS
    assert f(Q12, 8192, 50, post_layout=1) == N.BWD_FUSED
    assert f(Q12, 1024, 50, N.BWD_TWO_PASS, post_layout=1, inference=N.INF_LINEARIZE) == N.BWD_FUSED
    assert f(Q12, 1024, 50, post_layout=1, group_lanes=16) == N.BWD_FUSED
    assert f(Q12, 1024, 50, post_layout=1, group_lanes=-1) == -2  # no one-lane kernels for d = 16
    assert f(0, 4096, 200, post_layout=1) == -2               # trajectory-major posterior: wave-capable models only
    # ... and i2c_kernel_family answers from the same scalar fields
    fam = lambda *a, sweep=N.SWEEP_BACKWARD, **k: lib.i2c_kernel_family(ctypes.byref(_shape(*a, **k)), sweep)  # noqa: E731
    assert fam(Q12, 1024, 50, post_layout=1) == N.FAMILY_WAVE and fam(Q12, 8192, 50, post_layout=1) == N.FAMILY_QUAD
    assert fam(3, 4096, 300, sweep=N.SWEEP_FORWARD) == N.FAMILY_QUAD and fam(3, 4096, 300) == N.FAMILY_LANE
    # the quad propagation of the d = 16 model addresses a posterior / propagation cell through 32-bit offsets of one window, masked
    # stores parked at 2 GiB: the batch at which max(E_POST, E_PROP) B sizeof reaches 2 GiB is refused, not wrapped (round-5 advice)
    d = lib.query(Q12)
    e_prop = (d.nx + d.nu) + (d.nx + d.nu) * (d.nx + d.nu + 1) // 2 + d.nx + d.nx * (d.nx + 1) // 2
    b_max = (1 << 31) // (max(d.e_post, e_prop) * 8)
    assert fam(Q12, b_max, 50, post_layout=1, sweep=N.SWEEP_PROPAGATE) == N.FAMILY_QUAD
    assert fam(Q12, b_max + 1, 50, post_layout=1, sweep=N.SWEEP_PROPAGATE) == -1  # I2C_EINVAL


def test_bad_arguments_of_the_newer_entry_points():
    lib = pkg.load_library()
    N = pkg._native
    p = N.I2cProblem()
    p.abi_version = N.ABI_VERSION
    assert lib.i2c_mpc_step(ctypes.byref(p), None, None) == -1            # no step descriptor
    st = N.I2cMpcStep()
    assert lib.i2c_mpc_step(ctypes.byref(p), ctypes.byref(st), None) == -1  # null buffers
    assert lib.i2c_riccati_sweep(ctypes.byref(p), None, None, None, None, None, None, None) == -1
    assert lib.i2c_problem_size() == ctypes.sizeof(N.I2cProblem)
    p.inference = 7
    assert lib.i2c_forward_sweep(ctypes.byref(p), None, None, None, None, None) == -1


def test_design_family_table_is_current():
    """DESIGN.md section 8 ("what runs by default": model x batch window -> forward family / backward family + schedule / propagation
    / filter) is GENERATED from i2c_kernel_family / i2c_backward_schedule by tools/family_table.py; the committed block must be what
    the library answers today (round-5 review, item 6: a table that cannot drift)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("family_table", os.path.join(root, "tools", "family_table.py"))
    ft = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ft)
    doc = open(os.path.join(root, "DESIGN.md")).read()
    committed = doc[doc.index(ft.BEGIN): doc.index(ft.END) + len(ft.END)]
    assert committed == ft.block(pkg.load_library()), "DESIGN.md section 8 is stale: python tools/family_table.py --write"
