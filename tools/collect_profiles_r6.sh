#!/bin/bash
# Round-6 measurements on an MI355X box, into gpurun_out/r6p/ (summaries are then copied to profiles/r6_*):
#   * rocprofv3 --kernel-trace --stats of the default bench workload (pendulum T=200 B=4096); of the double cartpole T=300 B=4096,
#     the planar quadrotor and the cartpole with BOTH backward sweeps (the default chunked lane schedule and the round-6 quad walk,
#     group_lanes = 64); of the 12-state quadrotor at B = 1024 / 8192 / 32768;
#   * separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing) for the headline, the double cartpole and the
#     planar quadrotor -- the quad backward walk's traffic against its algorithmic bytes next to compose + stitch + walk;
#   * SQ / LDS / MFMA counters (tools/sq_counters.sh, four passes): headline, double cartpole (both backward sweeps), the CURRENT
#     k_quad_forward<Quadrotor12> at B = 32768 (the r4 file the bench line cited predates the square-root update), the control
#     steps of both quadrotors at B = 1024 (what the `issue` objects of the MPC legs cite).
# bench.py runs with --no-rccl under the profiler; the program after `--` is python3 itself (no launcher may re-exec).
# Usage (through gpurun):  bash tools/collect_profiles_r6.sh
set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r6p"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra --no-rccl"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r6_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
bash tools/sq_counters.sh r6_pendulum_B4096 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > "$OUT/sq_pendulum.txt" 2>&1
# d <= 8 models, default (chunked lane backward) AND group_lanes = 64 (quad backward walk) in one run: `wave`
for M in DoubleCartpoleKnown:300:dcp PlanarQuadrotor:50:planar CartpoleKnown:500:cartpole; do
  N=${M%%:*}; R=${M#*:}; T=${R%%:*}; S=${R##*:}
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_$S" -- python3 tools/bench_models.py f64 $N 4096 wave > "$OUT/${S}_4096.txt" 2> "$OUT/log_kt_$S.txt"
  if [ $S != cartpole ]; then
    rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_$S" -- python3 tools/bench_models.py f64 $N 4096 wave > /dev/null 2> "$OUT/log_f_$S.txt"
    rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_$S" -- python3 tools/bench_models.py f64 $N 4096 wave > /dev/null 2> "$OUT/log_w_$S.txt"
    python3 tools/pmc_summary.py "r6_${N}_B4096" 4096 $T "$OUT/f_$S" "$OUT/w_$S" > "$OUT/pmc_$S.txt"
  fi
  if [ $S = dcp ]; then
    bash tools/sq_counters.sh r6_${S}_B4096 tools/bench_models.py f64 $N 4096 wave > "$OUT/sq_$S.txt" 2>&1
  fi
done
# 12-state quadrotor: quad forward + quad backward (B = 8192, 32768), wave kernels (B = 1024); SQ counters of the current kernels
for B in 1024 8192 32768; do
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_$B" -- python3 tools/bench_models.py f64 Quadrotor12 $B > "$OUT/q12_$B.txt" 2> "$OUT/log_ktq_$B.txt"
done
bash tools/sq_counters.sh r6_quad12_B32768_quad_vs_wave tools/bench_models.py f64 Quadrotor12 32768 wave > "$OUT/sq_q12_32768.txt" 2>&1
bash tools/sq_counters.sh r6_quad12_mpc_B1024 tools/bench_mpc12.py 1024 > "$OUT/sq_q12_mpc.txt" 2>&1
bash tools/sq_counters.sh r6_planar_mpc_B1024 tools/bench_mpc.py 1024 > "$OUT/sq_planar_mpc.txt" 2>&1
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_planar_mpc" -- python3 tools/bench_mpc.py 1024 > "$OUT/planar_mpc_1024.txt" 2> "$OUT/log_kt_pm.txt"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_mpc" -- python3 tools/bench_mpc12.py 1024 > "$OUT/q12_mpc_1024.txt" 2> "$OUT/log_kt_qm.txt"
# plain timings
python3 tools/bench_quad_backward8.py mpc > "$OUT/quad_backward8_timings.txt" 2>&1
python3 tools/bench_mpc.py 1024 8192 > "$OUT/mpc_steps.txt" 2>&1
python3 tools/bench_mpc12.py 1024 8192 >> "$OUT/mpc_steps.txt" 2>&1
python3 tools/sweep_batch.py > "$OUT/batch_sweep.txt" 2>&1
python3 tools/bench_reference_shapes.py > "$OUT/reference_shapes.txt" 2>&1
# the quad passes of the chunked schedule at small batches (profiles/r6_quad_chunk_passes.txt; the A/B with the passes forced off:
# I2C_QUAD_PASSES_MAX_B=0), and every backward form in one kernel trace at B = 64 (profiles/r6_{dcp,planar}_B64_walkers_kernel_stats.csv)
python3 tools/bench_quad_backward8.py PlanarQuadrotor DoubleCartpoleKnown CartpoleKnown 1 64 128 256 512 768 1024 2048 4096 mpc > "$OUT/quad_chunk_passes.txt" 2>&1
I2C_QUAD_PASSES_MAX_B=0 python3 tools/bench_quad_backward8.py PlanarQuadrotor DoubleCartpoleKnown CartpoleKnown 1 64 256 768 1024 > "$OUT/quad_chunk_passes_off.txt" 2>&1
for M in DoubleCartpoleKnown:dcp PlanarQuadrotor:planar; do
  N=${M%%:*}; S=${M##*:}
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_${S}_B64" -- python3 tools/bench_quad_backward8.py $N 64 > "$OUT/${S}_B64.txt" 2> "$OUT/log_kt_${S}_B64.txt"
  f=$(find "$OUT/kt_${S}_B64" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r6_${S}_B64_walkers_kernel_stats.csv"
done
for d in kt kt_dcp kt_planar kt_cartpole kt_q12_1024 kt_q12_8192 kt_q12_32768 kt_planar_mpc kt_q12_mpc; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"
done
cp profiles/r6_*.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
