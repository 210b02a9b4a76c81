"""Builds (once) and loads the HOST SIMULATION of the kernel math: the csrc/ translation units compiled
by g++ with -DI2C_HOST_SIM, i.e. the very same templated cell code looped on the CPU.

Test infrastructure only: it lets `-m "not gpu"` tests check the kernels' arithmetic against
the oracle in a container without a GPU. The product package never builds, looks for, or
loads this library.
"""
import importlib
import importlib.util
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "input-inference-for-control_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_hostsim")
OUT = os.path.join(OUT_DIR, "libi2c_hostsim.so")


def build(force=False):
    """Rebuilds when a source is newer than the library. pytest-xdist workers arrive here together: an exclusive file lock makes
    one of them build and the others wait and find the library up to date."""
    import fcntl

    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "i2c_hip.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(ROOT, "input-inference-for-control_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    flags = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-DI2C_HOST_SIM"]
    return mod.compile_all("g++", flags, os.path.join(OUT_DIR, "obj"), OUT, ["-shared", "-fPIC"], verbose=False)


_lib = None


def load():
    global _lib
    if _lib is None:
        pkg = importlib.import_module("input-inference-for-control_amd")
        # I2C_HOSTSIM_LIB: a pre-built variant of the host simulation, e.g. the ASan / UBSan build of tools/sanitize.sh
        _lib = pkg.load_library(os.environ.get("I2C_HOSTSIM_LIB") or build())
        assert _lib.is_host_sim
    return _lib
