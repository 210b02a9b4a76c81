#!/bin/bash
# Round-3 measurements on an MI355X box, into gpurun_out/r3/ (summaries are then copied to profiles/r3_*):
#   * rocprofv3 --kernel-trace --stats of the default bench workload and of the 12-state quadrotor (wave and group kernels);
#   * separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing) for the headline and the quadrotor;
#   * SQ / LDS counters (tools/sq_counters.sh) of the quadrotor's wave and group kernels at B = 1024 and B = 4096;
#   * kernel stats + PMC of the saturated d >= 7 legs (double cartpole / quadrotor at B = 32768).
# Usage (through gpurun):  bash tools/collect_profiles_r3.sh
set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r3"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r3_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
# 12-state quadrotor: kernel stats of both families at the per-GPU share of config 4 (B = 1024) and at B = 4096
for B in 1024 4096; do
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_$B" -- python3 tools/bench_models.py f64 group wave Quadrotor12 $B > "$OUT/q12_$B.txt" 2> "$OUT/log_ktq_$B.txt"
done
# PMC traffic: wave kernels at B = 1024 and B = 4096 (explicitly requested), group kernels at B = 4096
for B in 1024 4096; do
  rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_q12w_$B" -- python3 tools/bench_models.py f64 wave Quadrotor12 $B > /dev/null 2> "$OUT/log_f_q12w_$B.txt"
  rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_q12w_$B" -- python3 tools/bench_models.py f64 wave Quadrotor12 $B > /dev/null 2> "$OUT/log_w_q12w_$B.txt"
  python3 tools/pmc_summary.py "r3_Quadrotor12_wave_B$B" $B 50 "$OUT/f_q12w_$B" "$OUT/w_q12w_$B" > "$OUT/pmc_q12w_$B.txt"
done
# SQ counters, both families
bash tools/sq_counters.sh r3_quad12_B1024_wave_vs_group tools/bench_models.py f64 group wave Quadrotor12 1024 > "$OUT/sq_quad12_1024.txt" 2>&1
bash tools/sq_counters.sh r3_quad12_B4096_wave_vs_group tools/bench_models.py f64 group wave Quadrotor12 4096 > "$OUT/sq_quad12_4096.txt" 2>&1
# the saturated d >= 7 legs of the bench line
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_dcp_32768" -- python3 tools/bench_models.py f64 DoubleCartpoleKnown 32768 > "$OUT/dcp_32768.txt" 2> "$OUT/log_ktd.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_dcp" -- python3 tools/bench_models.py f64 DoubleCartpoleKnown 32768 > /dev/null 2> "$OUT/log_f_dcp.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_dcp" -- python3 tools/bench_models.py f64 DoubleCartpoleKnown 32768 > /dev/null 2> "$OUT/log_w_dcp.txt"
python3 tools/pmc_summary.py "r3_DoubleCartpoleKnown_B32768" 32768 300 "$OUT/f_dcp" "$OUT/w_dcp" > "$OUT/pmc_dcp.txt"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_32768" -- python3 tools/bench_models.py f64 Quadrotor12 32768 > "$OUT/q12_32768.txt" 2> "$OUT/log_ktq32.txt"
cp profiles/r3_*.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
