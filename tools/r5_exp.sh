set -u
export TMPDIR=/tmp
O=gpurun_out/r5; mkdir -p $O
{
echo "== quad goldens"; python3 tools/quad_check.py
echo "== timings"; python3 tools/bench_models.py f64 DoubleCartpoleKnown CartpoleKnown PlanarQuadrotor 4096
python3 tools/bench_models.py f64 DoubleCartpoleKnown PlanarQuadrotor 8192
python3 tools/bench_models.py f64 Quadrotor12 4096 32768
} > $O/exp2.txt 2>&1
