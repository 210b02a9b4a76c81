"""In-tree build of the gfx950 library: hipcc cross-compiles without a GPU.

    python input-inference-for-control_amd/build.py [--force]
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include", "i2c_hip.h")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB = os.path.join(LIB_DIR, "libi2c_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [INCLUDE]


def build_hip(force=False, verbose=True):
    newest = max(os.path.getmtime(s) for s in sources())
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, "i2c_capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
