#!/bin/bash
# Round 5, review item 3b: chunked against fused backward schedule beyond the 256 MB Infinity Cache -- time (HIP events,
# tools/bench_models.py) and HBM traffic (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) at B = 8192 ... 32768 for the
# pendulum (T = 200) and the double cartpole (T = 300). Output: gpurun_out/r5/crossover/ (summary -> profiles/r5_backward_crossover.txt)
set -u
export TMPDIR=/tmp
O="$PWD/gpurun_out/r5/crossover"; rm -rf "$O"; mkdir -p "$O"
F="--output-format csv"
for M in PendulumKnown:200 DoubleCartpoleKnown:300; do
  N=${M%%:*}; T=${M##*:}
  for B in 8192 16384 24576 32768; do
    python3 tools/bench_models.py f64 $N $B chunked fused >> "$O/timings.txt" 2>/dev/null
    rocprofv3 --pmc FETCH_SIZE $F -d "$O/f_${N}_$B" -- python3 tools/bench_models.py f64 $N $B chunked fused > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE $F -d "$O/w_${N}_$B" -- python3 tools/bench_models.py f64 $N $B chunked fused > /dev/null 2>&1
    python3 tools/pmc_summary.py "r5x_${N}_B$B" $B $T "$O/f_${N}_$B" "$O/w_${N}_$B" >> "$O/traffic.txt" 2>&1
    cp "profiles/r5x_${N}_B${B}_pmc_traffic.json" "$O/" 2>/dev/null
    rm -rf "$O/f_${N}_$B" "$O/w_${N}_$B"
  done
done
