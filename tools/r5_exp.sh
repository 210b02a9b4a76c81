set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | grep -E "^E  .*Error|^FAILED|passed|failed" | head -40 > $O/exp5_tests.txt
python bench.py > $O/exp5_bench.json 2> $O/exp5_bench.err
