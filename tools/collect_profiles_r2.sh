#!/bin/bash
# Round-2 measurements on an MI355X box, into gpurun_out/r2/ (summaries are then copied to profiles/r2_*):
#   * rocprofv3 --kernel-trace --stats of the default bench workload, of the saturated batch, and of the other configs;
#   * separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing) for the headline and the d >= 7 models;
#   * SQ / LDS counters (tools/sq_counters.sh) of the pendulum (one lane vs 4 lanes per trajectory), the double cartpole
#     (one lane vs 16 lanes) and the 12-state quadrotor (16 lanes);
#   * the precision sweep at the full config shapes.
# Usage (through gpurun):  bash tools/collect_profiles_r2.sh
set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r2"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r2_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
# the other configs: kernel stats
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_models" -- python3 tools/bench_models.py f64 group DoubleCartpoleKnown PlanarQuadrotor Quadrotor12 PendulumKnown 4096 > "$OUT/models_4096.txt" 2> "$OUT/log_ktm.txt"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_8192" -- python3 tools/bench_models.py f64 Quadrotor12 8192 > "$OUT/q12_8192.txt" 2> "$OUT/log_ktq.txt"
# PMC traffic of the d >= 7 models (T differs per model: summarised per model)
for M in DoubleCartpoleKnown:300 Quadrotor12:50; do
  N=${M%%:*}; T=${M##*:}
  rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_$N" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_f_$N.txt"
  rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_$N" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_w_$N.txt"
  python3 tools/pmc_summary.py "r2_${N}_B4096" 4096 $T "$OUT/f_$N" "$OUT/w_$N" > "$OUT/pmc_$N.txt"
done
# SQ counters
bash tools/sq_counters.sh r2_pendulum_lane_vs_group tools/bench_models.py f64 group PendulumKnown 4096 > "$OUT/sq_pendulum.txt" 2>&1
bash tools/sq_counters.sh r2_dcp_lane_vs_group tools/bench_models.py f64 group DoubleCartpoleKnown 4096 > "$OUT/sq_dcp.txt" 2>&1
bash tools/sq_counters.sh r2_quad12 tools/bench_models.py f64 Quadrotor12 4096 > "$OUT/sq_quad12.txt" 2>&1
bash tools/sq_counters.sh r2_planar_quadrotor_lane_vs_group tools/bench_models.py f64 group PlanarQuadrotor 4096 > "$OUT/sq_planar.txt" 2>&1
python3 tools/precision_sweep.py "$OUT/precision_sweep.json" 4096 > "$OUT/precision.txt" 2>&1
python3 tools/bench_mpc.py 1 1024 8192 > "$OUT/mpc_planar.txt" 2>&1
cp profiles/r2_*.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/log_default.txt"
du -sh "$OUT"
