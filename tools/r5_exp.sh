set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5; mkdir -p $O
{
python3 tools/quad_check.py
for i in 1 2 3; do python3 tools/bench_models.py f64 DoubleCartpoleKnown CartpoleKnown PlanarQuadrotor 4096; done
python3 tools/bench_models.py f64 Quadrotor12 4096 32768
} > $O/exp8.txt 2>&1
