#!/bin/bash
# Round 6: chunk count of the chunked lane backward schedule at the BASELINE batches (I2C_CHUNKS: experiment knob of chunk_geometry,
# csrc/i2c_impl.hpp; read once per process). Default: min(65536 / B, 32, T / 4).
#   bash tools/r6_chunk_sweep.sh > gpurun_out/r6_chunk_sweep.txt
for spec in "PlanarQuadrotor 1024 0 8 10 12 13 17 25" "PlanarQuadrotor 4096 0 8 12 13 17" "DoubleCartpoleKnown 4096 0 8 12 16 20 25 30 38 50" "CartpoleKnown 4096 0 8 16 25 32 42" "PendulumKnown 4096 0 8 16 25 33 50"; do
  set -- $spec; M=$1; B=$2; shift 2
  for NC in "$@"; do
    echo "== $M B=$B I2C_CHUNKS=$NC (0 = default)"
    if [ "$NC" = 0 ]; then python3 tools/bench_models.py f64 $M $B 2>&1 | grep -v amdgpu; else I2C_CHUNKS=$NC python3 tools/bench_models.py f64 $M $B 2>&1 | grep -v amdgpu; fi
  done
done
