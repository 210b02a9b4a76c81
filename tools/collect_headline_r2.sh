set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r2b"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r2_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_models" -- python3 tools/bench_models.py f64 group DoubleCartpoleKnown PlanarQuadrotor Quadrotor12 PendulumKnown 4096 > "$OUT/models_4096.txt" 2> "$OUT/log_ktm.txt"
bash tools/sq_counters.sh r2_pendulum_lane_vs_group tools/bench_models.py f64 group PendulumKnown 4096 > "$OUT/sq_pendulum.txt" 2>&1
cp profiles/r2_B4096_pmc_traffic.json profiles/r2_pendulum_lane_vs_group_sq_counters.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/log_default.txt"
