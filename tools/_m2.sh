export TMPDIR=/tmp
mkdir -p gpurun_out/m2
python3 tools/quad_check.py em_quad12_T20 em_quad12_nondiag_T12 em_quad12_T12_propagate em_quad12_covctrl_T12 > gpurun_out/m2/check.txt 2>&1
python3 tools/bench_models.py f64 wave Quadrotor12 2048 4096 8192 32768 > gpurun_out/m2/q12.txt 2>&1
python3 tools/bench_mpc12.py 1024 8192 > gpurun_out/m2/mpc12.txt 2>&1
cat gpurun_out/m2/check.txt gpurun_out/m2/q12.txt gpurun_out/m2/mpc12.txt
