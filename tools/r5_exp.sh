set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
V=input-inference-for-control_amd/lib/variants
{
echo "== quad goldens"; python3 tools/quad_check.py
echo "== DCP with / without the scalar pre-elimination"
python3 tools/bench_models.py f64 DoubleCartpoleKnown 4096 8192
I2C_BENCH_LIB=$V/libi2c_hip_nolastlin.so python3 tools/bench_models.py f64 DoubleCartpoleKnown 4096 8192
python3 tools/bench_models.py f64 DoubleCartpoleKnown 4096
I2C_BENCH_LIB=$V/libi2c_hip_nolastlin.so python3 tools/bench_models.py f64 DoubleCartpoleKnown 4096
echo "== stamps"; I2C_BENCH_LIB=$V/libi2c_hip_stamps.so python3 tools/bench_models.py f64 DoubleCartpoleKnown 4096 | tail -3
echo "== crossover, cartpole and planar quadrotor"
for B in 8192 16384 24576 32768; do python3 tools/bench_models.py f64 CartpoleKnown PlanarQuadrotor $B chunked fused; done
} > $O/exp6.txt 2>&1
