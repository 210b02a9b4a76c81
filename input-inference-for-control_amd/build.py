"""In-tree build of the gfx950 library: hipcc cross-compiles without a GPU.

    python input-inference-for-control_amd/build.py [--force]

One translation unit per (model, dtype) pair (csrc/i2c_model_tu.hip compiled with -D flags) plus the C-ABI unit,
compiled in parallel and linked into lib/libi2c_hip.so. Objects go to build/ (git-ignored).
"""
import concurrent.futures
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include", "i2c_hip.h")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB = os.path.join(LIB_DIR, "libi2c_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# I2C_HIPCC_EXTRA: extra compiler flags for experiments, e.g. "-mllvm -amdgpu-sched-strategy=iterative-ilp" (measured:
# pendulum forward 298.6 -> 295.0 us, other models unchanged; not adopted)
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + os.environ.get("I2C_HIPCC_EXTRA", "").split()

# (struct in csrc/i2c_models.hpp, name used in csrc/i2c_entry.hpp); heaviest first so the pool stays busy
MODELS = [("Quadrotor12", "quadrotor12"), ("DoubleCartpole", "double_cartpole"), ("Quadrotor", "quadrotor"), ("Cartpole", "cartpole"),
          ("Pendulum", "pendulum"), ("PendulumActReg", "pendulum_actreg"), ("Linear", "linear"),
          ("LinearMinEnergy", "linear_minenergy")]
DTYPES = [("double", "f64", None), ("float", "f32", None), ("double", "f64s", "float")]  # (arithmetic, tag, storage)


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [INCLUDE]


def translation_units():
    """[(object name, source, extra -D flags)]"""
    tus = [(f"{name}_{tag}.o", "i2c_model_tu.hip",
            [f"-DI2C_TU_MODEL={struct}", f"-DI2C_TU_REAL={real}", f"-DI2C_TU_OPS=ops_{name}_{tag}"]
            + ([f"-DI2C_TU_STORE={store}"] if store else []))
           for struct, name in MODELS for real, tag, store in DTYPES]
    return tus + [("capi.o", "i2c_capi.hip", [])]


def compile_all(compiler, flags, obj_dir, lib, link_flags, verbose=True, jobs=None):
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    tus = translation_units()

    def one(tu):
        obj, src, defs = tu
        cmd = [compiler] + flags + defs + ["-c", os.path.join(CSRC, src), "-o", os.path.join(obj_dir, obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        return obj

    jobs = jobs or max(1, min(len(tus), (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(one, tus))
    cmd = [compiler] + link_flags + [os.path.join(obj_dir, o) for o in objs] + ["-o", lib]
    if verbose:
        print(f"{compiler}: {len(tus)} translation units ({jobs} at a time) -> {lib}", flush=True)
    subprocess.run(cmd, check=True)
    return lib


def build_hip(force=False, verbose=True):
    newest = max(os.path.getmtime(s) for s in sources())
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    return compile_all(HIPCC, FLAGS, OBJ_DIR, LIB, ["--offload-arch=gfx950", "-shared", "-fPIC"], verbose)


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
