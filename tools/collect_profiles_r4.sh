#!/bin/bash
# Round-4 measurements on an MI355X box, into gpurun_out/r4/ (summaries are then copied to profiles/r4_*):
#   * rocprofv3 --kernel-trace --stats of the default bench workload (pendulum T=200 B=4096), of the double cartpole T=300
#     B=4096 and the planar quadrotor T=50 B=4096 (quad forward kernel), of the 12-state quadrotor at B = 8192 / 32768 (quad
#     forward + wave backward) and B = 1024 (wave kernels);
#   * separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing) for the headline, the double cartpole
#     (B = 4096) and the 12-state quadrotor (B = 8192);
#   * SQ / LDS / MFMA counters (tools/sq_counters.sh, four passes) of the same kernels.
# bench.py runs with --no-rccl under the profiler (no process group, no RCCL kernels in the traces).
# Usage (through gpurun):  bash tools/collect_profiles_r4.sh
set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r4"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra --no-rccl"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r4_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
bash tools/sq_counters.sh r4_pendulum_B4096 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > "$OUT/sq_pendulum.txt" 2>&1
# d <= 8 models on the quad forward kernel
for M in DoubleCartpoleKnown:300:dcp PlanarQuadrotor:50:planar CartpoleKnown:500:cartpole; do
  N=${M%%:*}; R=${M#*:}; T=${R%%:*}; S=${R##*:}
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_$S" -- python3 tools/bench_models.py f64 $N 4096 > "$OUT/${S}_4096.txt" 2> "$OUT/log_kt_$S.txt"
  rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_$S" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_f_$S.txt"
  rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_$S" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_w_$S.txt"
  python3 tools/pmc_summary.py "r4_${N}_B4096" 4096 $T "$OUT/f_$S" "$OUT/w_$S" > "$OUT/pmc_$S.txt"
  bash tools/sq_counters.sh r4_${S}_B4096 tools/bench_models.py f64 $N 4096 > "$OUT/sq_$S.txt" 2>&1
done
# the double cartpole on the families the quad kernel replaced (same box, same run): group forward (round 2/3 default), lane forward
bash tools/sq_counters.sh r4_dcp_B4096_group_and_lane tools/bench_models.py f64 DoubleCartpoleKnown 4096 group lane > "$OUT/sq_dcp_old.txt" 2>&1
# 12-state quadrotor: quad forward + wave backward (default from B = 2048), wave kernels throughout (64)
for B in 1024 8192 32768; do
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_$B" -- python3 tools/bench_models.py f64 wave Quadrotor12 $B > "$OUT/q12_$B.txt" 2> "$OUT/log_ktq_$B.txt"
done
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_q12" -- python3 tools/bench_models.py f64 Quadrotor12 8192 > /dev/null 2> "$OUT/log_f_q12.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_q12" -- python3 tools/bench_models.py f64 Quadrotor12 8192 > /dev/null 2> "$OUT/log_w_q12.txt"
python3 tools/pmc_summary.py "r4_Quadrotor12_B8192" 8192 50 "$OUT/f_q12" "$OUT/w_q12" > "$OUT/pmc_q12.txt"
bash tools/sq_counters.sh r4_quad12_B8192_quad_vs_wave tools/bench_models.py f64 wave Quadrotor12 8192 > "$OUT/sq_q12_8192.txt" 2>&1
bash tools/sq_counters.sh r4_quad12_B32768_quad_vs_wave tools/bench_models.py f64 wave Quadrotor12 32768 > "$OUT/sq_q12_32768.txt" 2>&1
# closed-loop control steps (planar quadrotor, one EM iteration per step; 12-state quadrotor, two)
python3 tools/bench_mpc.py 1024 8192 > "$OUT/mpc_steps.txt" 2>&1
python3 tools/bench_mpc12.py 1024 8192 >> "$OUT/mpc_steps.txt" 2>&1
for d in kt kt_dcp kt_planar kt_cartpole kt_q12_1024 kt_q12_8192 kt_q12_32768; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"
  python3 tools/kstats.py "$OUT/$d" > "$OUT/${d}_kstats.txt" 2>/dev/null
done
cp profiles/r4_*.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
