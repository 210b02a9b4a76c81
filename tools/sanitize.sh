#!/bin/bash
# AddressSanitizer + UBSan run of the kernel math: the host simulation (csrc/ compiled by g++ -DI2C_HOST_SIM) rebuilt with
# -fsanitize=address,undefined and driven by the CPU test-suite. (GPU sanitizers are not available on the pool.)
set -eu
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/i2c_asan}
python3 - "$OUT" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("b", "input-inference-for-control_amd/build.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-DI2C_HOST_SIM",
         "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
m.compile_all("g++", flags, sys.argv[1] + "/obj", sys.argv[1] + "/libi2c_hostsim.so",
              ["-shared", "-fPIC", "-fsanitize=address,undefined"])
PY
I2C_HOSTSIM_LIB="$OUT/libi2c_hostsim.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_kernels_hostsim.py tests/test_edge_cases.py tests/test_mpc.py \
  tests/test_feature_matrix.py tests/test_rollout.py tests/test_lqr_known_answer.py tests/test_facade.py tests/test_dist_gloo.py -q -m "not gpu" 2>&1 | tee "$OUT/report.txt" | tail -3
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$OUT/report.txt" || true)"
