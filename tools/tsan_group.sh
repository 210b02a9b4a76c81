#!/bin/bash
# ThreadSanitizer run of the group kernels' host simulation (see tools/tsan_group.cpp). Usage: bash tools/tsan_group.sh [outdir]
set -eu
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/i2c_tsan}
mkdir -p "$OUT"
python3 - "$OUT" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("b", "input-inference-for-control_amd/build.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-DI2C_HOST_SIM", "-fsanitize=thread"]
m.compile_all("g++", flags, sys.argv[1] + "/obj", sys.argv[1] + "/libi2c_hostsim_tsan.so", ["-shared", "-fPIC", "-fsanitize=thread"], verbose=False)
PY
g++ -O1 -g -std=c++17 -fsanitize=thread tools/tsan_group.cpp -o "$OUT/tsan_group" -L"$OUT" -li2c_hostsim_tsan -Wl,-rpath,"$OUT" -pthread
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" "$OUT/tsan_group" 2>&1 | tee "$OUT/report.txt" | tail -5
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' "$OUT/report.txt" || true)"
