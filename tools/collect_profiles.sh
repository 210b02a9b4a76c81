#!/bin/bash
# Collects the judged measurements on an MI355X box into gpurun_out/<tag>/ :
#   rocprofv3 --kernel-trace --stats of the default bench workload and of the saturated batch,
#   separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing), and the default bench line.
# Usage (through gpurun):  bash tools/collect_profiles.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-profiles_run}
OUT="$PWD/gpurun_out/$TAG"
mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated > /dev/null 2> "$OUT/log_w.txt"
BIG="--batch 131072 --no-cpu-baseline --no-saturated"
rocprofv3 --kernel-trace --stats $F -d "$OUT/ktbig" -- python3 bench.py $BIG --steps 20 --warmup 2 > "$OUT/bench_ktbig.json" 2> "$OUT/log_ktbig.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/fbig" -- python3 bench.py $BIG --steps 10 --warmup 2 > /dev/null 2> "$OUT/log_fb.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/wbig" -- python3 bench.py $BIG --steps 10 --warmup 2 > /dev/null 2> "$OUT/log_wb.txt"
python3 tools/pmc_summary.py "tmp_${TAG}_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > /dev/null
python3 tools/pmc_summary.py "tmp_${TAG}_B131072" 131072 200 "$OUT/fbig" "$OUT/wbig" > /dev/null
cp profiles/tmp_${TAG}_*_pmc_traffic.json "$OUT/"
find "$OUT" -name "*.csv" -size +3M -delete
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/log_default.txt"
du -sh "$OUT"
