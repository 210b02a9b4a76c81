export TMPDIR=/tmp
mkdir -p gpurun_out/m1
python3 tools/bench_models.py f64 PendulumKnown 1024 4096 8192 wave > gpurun_out/m1/pend.txt 2>&1
python3 tools/bench_models.py f64 CartpoleKnown 4096 wave >> gpurun_out/m1/pend.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m1/kt_mpc -- python3 tools/bench_mpc.py 1024 > gpurun_out/m1/mpc.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m1/kt_mpc12 -- python3 tools/bench_mpc12.py 1024 > gpurun_out/m1/mpc12.txt 2>&1
for d in kt_mpc kt_mpc12; do python3 tools/kstats.py gpurun_out/m1/$d > gpurun_out/m1/${d}_kstats.txt 2>/dev/null; done
find gpurun_out/m1 -name "*.csv" -size +1M -delete; find gpurun_out/m1 -name "*.db" -delete
cat gpurun_out/m1/pend.txt gpurun_out/m1/mpc.txt gpurun_out/m1/mpc12.txt gpurun_out/m1/*kstats.txt
