"""Build a second copy of the gfx950 library with extra -D flags, for same-box A/B measurements:
python tools/build_variant.py <tag> -DNAME=VALUE ...   ->  input-inference-for-control_amd/build/variants/libi2c_hip_<tag>.so
(select it with I2C_BENCH_LIB=<path> python tools/bench_models.py ...)"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(ROOT, "input-inference-for-control_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
tag, defs = sys.argv[1], sys.argv[2:]
lib = os.path.join(b.OBJ_DIR, "variants", f"libi2c_hip_{tag}.so")
print(b.compile_all(b.HIPCC, b.FLAGS + defs, os.path.join(b.OBJ_DIR, "var_" + tag), lib, ["--offload-arch=gfx950", "-shared", "-fPIC"]))
