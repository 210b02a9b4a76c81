"""Opcode-class histogram of the main loop of a kernel, from the assembly hipcc emits for a (model, dtype) translation unit.

    python tools/isa_histogram.py <Model struct> <ops name> <kernel name substring> [<mangled-name regex>]

e.g.  python tools/isa_histogram.py Pendulum pendulum k_forward 'k_forwardINS_8PendulumEdLb1ELb0EdEE'   (the lean sigma-point forward sweep)
Compiles csrc/i2c_model_tu.hip exactly as build.py does (plus -save-temps), finds the kernel, takes its largest loop (the time
loop of a sweep) and counts instructions by class. Static counts of ONE loop body: run-time branches inside it (feed-forward
/ feedback prior, the terminal cell) are both counted."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = [
    ("v_mfma", "matrix (v_mfma_f64_*)"),
    ("v_fma_f64|v_fmac_f64", "fp64 fma"), ("v_mul_f64", "fp64 mul"), ("v_add_f64", "fp64 add"),
    ("v_rsq_f64|v_rcp_f64|v_sqrt_f64", "fp64 rsq / rcp (quarter rate)"), ("v_ldexp_f64|v_rndne_f64|v_cvt_.*f64|v_frexp|v_trig|v_fract_f64", "fp64 cvt / ldexp / rndne"),
    ("v_max_f64|v_min_f64|v_cmp_.*_f64|v_cmpx_.*f64|v_cmp_class_f64", "fp64 min / max / compare"),
    ("v_cndmask", "select (v_cndmask_b32)"), ("v_mov_b32_dpp|v_mov_b64_dpp|.*_dpp", "DPP moves"),
    ("v_mov_b32|v_mov_b64|v_accvgpr", "moves (v_mov, accvgpr)"), ("v_readlane|v_readfirstlane|v_writelane|v_permlane", "lane <-> scalar / permlane"),
    ("v_", "other vector (integer, address, logic)"),
    ("s_waitcnt", "s_waitcnt"), ("s_nop", "s_nop"), ("s_cbranch|s_branch|s_setpc|s_endpgm", "branches"), ("s_load|s_buffer_load", "scalar loads"),
    ("s_", "other scalar"),
    ("ds_", "LDS"), ("buffer_load|global_load|flat_load|scratch_load", "vector memory loads"), ("buffer_store|global_store|flat_store|scratch_store", "vector memory stores"),
]


def main():
    model, name, kern = sys.argv[1], sys.argv[2], sys.argv[3]
    pat = sys.argv[4] if len(sys.argv) > 4 else kern
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", f"-DI2C_TU_MODEL={model}", "-DI2C_TU_REAL=double",
               f"-DI2C_TU_OPS=ops_{name}_f64", "-c", os.path.join(ROOT, "input-inference-for-control_amd", "csrc", "i2c_model_tu.hip"),
               "-o", os.path.join(d, "x.o"), "-save-temps=obj"]
        subprocess.run(cmd, check=True, capture_output=True, cwd=d)
        lines = open(os.path.join(d, "i2c_model_tu-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i)
    ops = collections.Counter()
    for l in body[best[0]: best[1] + 1]:
        l = l.strip()
        if l and not l.startswith((";", ".")) and not l.endswith(":"):
            ops[l.split()[0]] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        cls[next(label for rx, label in CLASSES if re.match(rx, op))] += n
    print(f"{lines[start].split(':')[0]}")
    print(f"main loop: {best[1] - best[0] + 1} lines, {sum(ops.values())} instructions (static, one loop body)")
    for _, label in CLASSES:
        if cls[label]:
            print(f"  {label:44s} {cls[label]:5d}")
    vec = sum(n for op, n in ops.items() if op.startswith("v_"))
    print(f"  vector instructions {vec}, of them fp64 arithmetic {sum(n for op, n in ops.items() if re.match(r'v_(fma|fmac|mul|add|rsq|rcp|sqrt|max|min|ldexp|rndne)_f64', op))}")
    print("top opcodes: " + ", ".join(f"{op} {n}" for op, n in ops.most_common(24)))


if __name__ == "__main__":
    main()
