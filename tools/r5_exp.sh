set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | grep -E "^E  .*Error|^FAILED|passed|failed" | head -40 > $O/exp9_tests.txt
python3 tools/bench_mpc12.py 1024 8192 > $O/exp9_mpc.txt 2>&1
python3 tools/bench_mpc.py 1024 8192 >> $O/exp9_mpc.txt 2>&1
python3 tools/fuzz_gpu.py 2000 300 > $O/exp9_fuzz.txt 2>&1
