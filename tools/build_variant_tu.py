"""Build a variant of the gfx950 library in which only SOME (model, dtype) translation units are recompiled with extra flags; the
other objects are taken from the standard build (input-inference-for-control_amd/build/, which must be current). A one-model
variant takes about a minute instead of the several of tools/build_variant.py.

    python tools/build_variant_tu.py <tag> <name>_<dtype>[,<name>_<dtype>...] [flags ...]
    e.g. python tools/build_variant_tu.py ilp double_cartpole_f64 -mllvm -amdgpu-sched-strategy=max-ilp
    ->   input-inference-for-control_amd/lib/variants/libi2c_hip_<tag>.so   (select it with I2C_BENCH_LIB=<path>)
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(ROOT, "input-inference-for-control_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)


def main():
    tag, which, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    b.build_hip(verbose=False)  # the standard objects
    var_dir = os.path.join(b.OBJ_DIR, "var_" + tag)
    os.makedirs(var_dir, exist_ok=True)
    objs, procs = [], []
    for obj, src, defs in b.translation_units():
        if obj[:-2] in which:
            out = os.path.join(var_dir, obj)
            cmd = [b.HIPCC] + b.FLAGS + flags + defs + ["-c", os.path.join(b.CSRC, src), "-o", out]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            objs.append(out)
        else:
            objs.append(os.path.join(b.OBJ_DIR, obj))
    assert len(procs) == len(which), f"unknown translation unit in {which}"
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(" ".join(cmd) + "\n" + out)
    lib = os.path.join(b.LIB_DIR, "variants", f"libi2c_hip_{tag}.so")  # (lib/ travels to the GPU box with gpurun, build/ does not)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
