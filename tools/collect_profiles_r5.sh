#!/bin/bash
# Round-5 measurements on an MI355X box, into gpurun_out/r5p/ (summaries are then copied to profiles/r5_*):
#   * rocprofv3 --kernel-trace --stats of the default bench workload (pendulum T=200 B=4096) and of the double cartpole T=300
#     B=4096 (quad forward kernel with the round-5 pivot look-ahead and the scalar pre-elimination of its 9th observation),
#     the Gauss-Hermite rule on the headline shape, the planar quadrotor and the cartpole;
#   * separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with tracing) for the headline and the double cartpole;
#   * SQ / LDS / MFMA counters (tools/sq_counters.sh, four passes) of the same kernels.
# bench.py runs with --no-rccl under the profiler (no process group, no RCCL kernels in the traces); the program after `--` is
# python3 itself (the profiler's preloaded library has initialised the GPU: no launcher may re-exec).
# Usage (through gpurun):  bash tools/collect_profiles_r5.sh
set -u
export TMPDIR=/tmp
OUT="$PWD/gpurun_out/r5p"
rm -rf "$OUT"; mkdir -p "$OUT"
F="--output-format csv"
SMALL="--steps 50 --warmup 5 --no-cpu-baseline --no-saturated --no-extra --no-rccl"
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt" -- python3 bench.py $SMALL > "$OUT/bench_kt.json" 2> "$OUT/log_kt.txt"
rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_f.txt"
rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w4096" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > /dev/null 2> "$OUT/log_w.txt"
python3 tools/pmc_summary.py "r5_B4096" 4096 200 "$OUT/f4096" "$OUT/w4096" > "$OUT/pmc_B4096.txt"
bash tools/sq_counters.sh r5_pendulum_B4096 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-saturated --no-extra --no-rccl > "$OUT/sq_pendulum.txt" 2>&1
for M in DoubleCartpoleKnown:300:dcp PlanarQuadrotor:50:planar CartpoleKnown:500:cartpole; do
  N=${M%%:*}; R=${M#*:}; T=${R%%:*}; S=${R##*:}
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_$S" -- python3 tools/bench_models.py f64 $N 4096 > "$OUT/${S}_4096.txt" 2> "$OUT/log_kt_$S.txt"
  if [ $S = dcp ]; then
    rocprofv3 --pmc FETCH_SIZE $F -d "$OUT/f_$S" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_f_$S.txt"
    rocprofv3 --pmc WRITE_SIZE $F -d "$OUT/w_$S" -- python3 tools/bench_models.py f64 $N 4096 > /dev/null 2> "$OUT/log_w_$S.txt"
    python3 tools/pmc_summary.py "r5_${N}_B4096" 4096 $T "$OUT/f_$S" "$OUT/w_$S" > "$OUT/pmc_$S.txt"
  fi
  bash tools/sq_counters.sh r5_${S}_B4096 tools/bench_models.py f64 $N 4096 > "$OUT/sq_$S.txt" 2>&1
done
# the other inference rules on the headline shape (Gauss-Hermite: unrolled grid + chunked backward)
rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_inf" -- python3 tools/bench_inference.py > "$OUT/inference_4096.txt" 2> "$OUT/log_kt_inf.txt"
# 12-state quadrotor: quad forward + quad backward (B = 8192, 32768), wave kernels (B = 1024)
for B in 1024 8192 32768; do
  rocprofv3 --kernel-trace --stats $F -d "$OUT/kt_q12_$B" -- python3 tools/bench_models.py f64 Quadrotor12 $B > "$OUT/q12_$B.txt" 2> "$OUT/log_ktq_$B.txt"
done
python3 tools/bench_mpc.py 1024 8192 > "$OUT/mpc_steps.txt" 2>&1
python3 tools/bench_mpc12.py 1024 8192 >> "$OUT/mpc_steps.txt" 2>&1
python3 tools/sweep_batch.py > "$OUT/batch_sweep.txt" 2>&1
for d in kt kt_dcp kt_planar kt_cartpole kt_inf kt_q12_1024 kt_q12_8192 kt_q12_32768; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"
done
cp profiles/r5_*.json "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
