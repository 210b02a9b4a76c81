"""In-tree build of the gfx950 library: hipcc cross-compiles without a GPU.

    python input-inference-for-control_amd/build.py [--force]
    python input-inference-for-control_amd/build.py --model path/to/my_model.hpp [--struct MyModel] [--name my_model]
        -> lib/libi2c_model_<name>.so: an OUT-OF-TREE model (a functor struct derived from i2c::ModelDefaults, see
           csrc/i2c_models.hpp and INTEGRATION.md section 3) compiled against the same kernels, without touching the tree;
           load it with NativeLibrary.load_model / i2c_load_model (include/i2c_hip.h).

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


HOST_SIM_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-DI2C_HOST_SIM"]  # (tests/hostsim.py)


def model_lib_path(name, host_sim=False, out_dir=None):
    return os.path.join(out_dir or LIB_DIR, f"libi2c_model_{name}{'_hostsim' if host_sim else ''}.so")


def build_model(header, struct=None, name=None, host_sim=False, out_dir=None, force=False, verbose=True):
    """Compile ONE out-of-tree model header into a model library: csrc/i2c_model_tu.hip once per precision with the header
    included (-DI2C_TU_HEADER) plus csrc/i2c_model_entry.hip (the C symbols i2c_load_model resolves). `struct` = the functor's
    name in namespace i2c (default: the header's file name in CamelCase), `name` = the library's name (default: the file name).
    host_sim=True builds the CPU simulation of the same kernels with g++ (tests only)."""
    header = os.path.abspath(header)
    stem = os.path.splitext(os.path.basename(header))[0]
    name = name or stem
    struct = struct or "".join(w.capitalize() for w in stem.split("_"))
    lib = model_lib_path(name, host_sim, out_dir)
    newest = max(os.path.getmtime(s) for s in sources() + [header])
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= newest:
        return lib
    compiler, flags, link = (("g++", HOST_SIM_FLAGS, ["-shared", "-fPIC", "-pthread"]) if host_sim else
                             (HIPCC, FLAGS, ["--offload-arch=gfx950", "-shared", "-fPIC"]))
    obj_dir = os.path.join(OBJ_DIR, f"model_{name}{'_hostsim' if host_sim else ''}")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    tus = [(f"{name}_{tag}.o", "i2c_model_tu.hip",
            [f'-DI2C_TU_HEADER="{header}"', f"-DI2C_TU_MODEL={struct}", f"-DI2C_TU_REAL={real}", f"-DI2C_TU_OPS=ops_{name}_{tag}"]
            + ([f"-DI2C_TU_STORE={store}"] if store else []))
           for real, tag, store in DTYPES] + [("entry.o", "i2c_model_entry.hip", [f"-DI2C_PLUGIN_NAME={name}"])]

    def one(tu):
        obj, src, defs = tu
        cmd = [compiler] + flags + defs + ["-c", os.path.join(CSRC, src), "-o", os.path.join(obj_dir, obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        return os.path.join(obj_dir, obj)

    with concurrent.futures.ThreadPoolExecutor(len(tus)) as pool:
        objs = list(pool.map(one, tus))
    if verbose:
        print(f"{compiler}: model {struct} ({header}) -> {lib}", flush=True)
    subprocess.run([compiler] + link + objs + ["-o", lib], check=True)
    return lib


if __name__ == "__main__":
    def opt(flag):
        return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv else None

    if "--model" in sys.argv:
        print(build_model(opt("--model"), struct=opt("--struct"), name=opt("--name"), host_sim="--host-sim" in sys.argv,
                          force="--force" in sys.argv))
    else:
        print(build_hip(force="--force" in sys.argv))
