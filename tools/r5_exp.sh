set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | grep -E "^E  .*Error|^FAILED|passed|failed" | head -40 > $O/exp7_tests.txt
python bench.py > $O/exp7_bench.json 2> $O/exp7_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/exp7_smoke.txt 2>&1
