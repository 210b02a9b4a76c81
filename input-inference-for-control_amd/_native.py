"""ctypes binding of the C ABI in include/i2c_hip.h.

The product path loads exactly one library: ``lib/libi2c_hip.so`` (hipcc, gfx950), built
in-tree by ``__graft_entry__.build()`` / ``build.py``. If it is missing or does not export the
whole ABI, loading fails loudly -- there is no CPU fallback. (tests/ may hand an explicitly
built host-simulation library to :func:`load_library` to check kernel numerics on a box
without a GPU; nothing in the package ever looks for it.)
"""
import ctypes as C
import os

ABI_VERSION = 7
MAX_NX, MAX_NU, MAX_NZ, MAX_PARAMS = 12, 4, 16, 16
MAX_GH_DEGREE = 8


def _sym(n):
    return n * (n + 1) // 2


F64, F32, F64_F32S = 0, 1, 2  # I2cProblem.dtype (include/i2c_hip.h): F64_F32S = fp64 arithmetic on fp32-stored per-cell buffers
BWD_AUTO, BWD_TWO_PASS, BWD_FUSED, BWD_CHUNKED = 0, 1, 2, 3
INF_CUBATURE, INF_LINEARIZE, INF_GAUSS_HERMITE = 0, 1, 2
FAMILY_LANE, FAMILY_GROUP, FAMILY_WAVE, FAMILY_QUAD = 1, 2, 3, 4  # i2c_kernel_family()
FAMILY_NAMES = {FAMILY_LANE: "lane", FAMILY_GROUP: "group", FAMILY_WAVE: "wave", FAMILY_QUAD: "quad"}
SWEEP_FORWARD, SWEEP_BACKWARD, SWEEP_PROPAGATE, SWEEP_FILTER, SWEEP_CHUNK_PASSES, SWEEP_CHUNK_STITCH = 0, 1, 2, 3, 4, 5
LANES_QUAD = 164  # I2cProblem.group_lanes: the quad kernels of a model that also has wave kernels; on a d <= 8 model the quad FORWARD sweep only (I2C_LANES_QUAD)

PLUGIN_BASE = 64  # I2C_MODEL_PLUGIN_BASE: ids of out-of-tree models start here

MODEL_IDS = {
    "PendulumKnown": 0,
    "PendulumKnownActReg": 1,
    "CartpoleKnown": 2,
    "DoubleCartpoleKnown": 3,
    "LinearKnown": 4,
    "LinearKnownMinimumEnergy": 5,
    "PlanarQuadrotor": 6,
    "Quadrotor12": 7,
}

FAIL_REASONS = {
    1: "prior joint covariance sig_xu0_f is not positive definite",
    2: "pdf-ratio covariance sig_xx + sig_x0_f is not positive definite",
    3: "cost-observation covariance sig_z0_f + sig_xi is not positive definite",
    4: "updated joint covariance sig_xu1_f is not positive definite",
    5: "predicted state covariance sig_x3_f is not positive definite",
    6: "terminal observation / terminal prior update failed",
    7: "posterior joint covariance sig_xu0_m is not positive definite",
    8: "closed-loop propagation covariance is not positive definite",
    9: "cubature Kalman filter covariance is not positive definite",
    10: "improper backward message in the Riccati form (a matrix to invert is not positive definite)",
}


class I2cDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "nu", "nz", "nzt", "e_post", "e_fwd", "e_xm", "e_zpost", "e_prop", "n_params", "ny",
                                          "group_lanes", "group_only", "wave", "quad")]


class I2cProblem(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("model_id", C.c_int32),
        ("dtype", C.c_int32),
        ("B", C.c_int32),
        ("T", C.c_int32),
        ("has_Qf", C.c_int32),
        ("has_x_terminal", C.c_int32),
        ("z_per_cell", C.c_int32),
        ("backward_mode", C.c_int32),
        ("terminal_cell", C.c_int32),
        ("inference", C.c_int32),
        ("expert_controller", C.c_int32),
        ("gh_degree", C.c_int32),
        ("group_lanes", C.c_int32),
        ("t0", C.c_int32),
        ("post_layout", C.c_int32),
        ("quad_alpha", C.c_double),
        ("quad_beta", C.c_double),
        ("quad_kappa", C.c_double),
        ("dtemp", C.c_double),
        ("gh_nodes", C.c_double * MAX_GH_DEGREE),
        ("gh_weights", C.c_double * MAX_GH_DEGREE),
        ("sig_eta", C.c_double * _sym(MAX_NX)),
        ("sig_xi0", C.c_double * _sym(MAX_NZ)),
        ("QR", C.c_double * _sym(MAX_NZ)),
        ("sig_xiT0", C.c_double * _sym(MAX_NZ)),
        ("Qf", C.c_double * _sym(MAX_NZ)),
        ("zg", C.c_double * MAX_NZ),
        ("zg_term", C.c_double * MAX_NZ),
        ("mu_x_term", C.c_double * MAX_NX),
        ("sig_x_term", C.c_double * _sym(MAX_NX)),
        ("model_params", C.c_double * MAX_PARAMS),
        ("x0", C.c_void_p),
        ("sig_x0", C.c_void_p),
        ("z", C.c_void_p),
        ("alpha", C.c_void_p),
        ("alpha_cell", C.c_void_p),
        ("temp", C.c_void_p),
        ("work", C.c_void_p),
        ("feedforward", C.c_void_p),
        ("expert", C.c_void_p),
    ]


class I2cMpcStep(C.Structure):
    _fields_ = [
        ("do_filter", C.c_int32), ("n_iter", C.c_int32), ("tau", C.c_int32), ("reserved0", C.c_int32),
        ("sig_zeta", C.c_double * _sym(MAX_NZ)),
        ("y", C.c_void_p), ("u", C.c_void_p), ("post", C.c_void_p), ("fwd", C.c_void_p),
        ("xm", C.c_void_p), ("zpost", C.c_void_p), ("cell_stats", C.c_void_p), ("term_stats", C.c_void_p),
        ("cell_init", C.c_void_p), ("alpha_init", C.c_void_p), ("z_new", C.c_void_p),
        ("action", C.c_void_p), ("status", C.c_void_p),
    ]


_SIGNATURES = {
    "i2c_abi_version": (C.c_int, []),
    "i2c_problem_size": (C.c_size_t, []),
    "i2c_build_info": (C.c_char_p, []),
    "i2c_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "i2c_backward_schedule": (C.c_int, [C.POINTER(I2cProblem)]),
    "i2c_query": (C.c_int, [C.c_int, C.POINTER(I2cDims)]),
    "i2c_register_model": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(I2cDims)]),
    "i2c_load_model": (C.c_int, [C.c_char_p, C.POINTER(I2cDims)]),
    "i2c_kernel_family": (C.c_int, [C.POINTER(I2cProblem), C.c_int]),
    "i2c_forward_sweep": (C.c_int, [C.POINTER(I2cProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "i2c_backward_sweep": (
        C.c_int,
        [C.POINTER(I2cProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "i2c_mstep": (C.c_int, [C.POINTER(I2cProblem), C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "i2c_learn": (
        C.c_int,
        [C.POINTER(I2cProblem)] + [C.c_void_p] * 6 + [C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "i2c_learn_propagate": (
        C.c_int,
        [C.POINTER(I2cProblem)] + [C.c_void_p] * 8 + [C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "i2c_riccati_sweep": (C.c_int, [C.POINTER(I2cProblem)] + [C.c_void_p] * 7),
    "i2c_mpc_step": (C.c_int, [C.POINTER(I2cProblem), C.POINTER(I2cMpcStep), C.c_void_p]),
    "i2c_shift_horizon": (C.c_int, [C.POINTER(I2cProblem)] + [C.c_void_p] * 6),
    "i2c_rollout": (C.c_int, [C.POINTER(I2cProblem), C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "i2c_ckf_filter": (
        C.c_int,
        [C.POINTER(I2cProblem), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "i2c_propagate": (C.c_int, [C.POINTER(I2cProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(PKG_DIR, "lib", "libi2c_hip.so")


class NativeLibrary:
    """A loaded i2c C-ABI library with typed entry points."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                f"i2c HIP library not found at {path}; build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        self.path = path
        self._dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise ImportError(f"{path} does not export {name} (include/i2c_hip.h)") from e
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        if self.i2c_abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {self.i2c_abi_version()} != {ABI_VERSION}")
        if self.i2c_problem_size() != C.sizeof(I2cProblem):
            raise ImportError(f"{path}: sizeof(I2cProblem) = {self.i2c_problem_size()} but the ctypes mirror has "
                              f"{C.sizeof(I2cProblem)} bytes (include/i2c_hip.h and _native.py disagree)")
        self.build_info = self.i2c_build_info().decode()
        self.is_host_sim = "host-simulation" in self.build_info

    def load_model(self, path):
        """Register an out-of-tree model library (built by build.py's build_model) with this library: -> (model_id, I2cDims)."""
        if not os.path.exists(path):
            raise ImportError(f"model library not found at {path}; build it with `python build.py --model <header>`")
        d = I2cDims()
        mid = self.i2c_load_model(os.fsencode(path), C.byref(d))
        if mid < 0:
            raise ImportError(f"i2c_load_model({path}) failed with {mid}: not a model library of ABI version {ABI_VERSION}, or its "
                              "dimensions exceed the capacities of I2cProblem (include/i2c_hip.h)")
        return mid, d

    def query(self, model_id):
        d = I2cDims()
        rc = self.i2c_query(int(model_id), C.byref(d))
        if rc != 0:
            raise ValueError(f"i2c_query({model_id}) failed with {rc}")
        return d


_default = None


def load_library(path=None):
    """Load (once) the gfx950 library; `path` is only for tests that pass an explicit build."""
    global _default
    if path is not None:
        return NativeLibrary(path)
    if _default is None:
        _default = NativeLibrary(DEFAULT_LIB)
    return _default
