#!/bin/bash
# SQ / LDS counters of every i2c kernel a command launches: separate rocprofv3 --pmc passes (never combined with tracing,
# at most 8 SQ counters per pass), summarised per kernel by tools/sq_summary.py.
# Usage (through gpurun):  bash tools/sq_counters.sh <tag> <python script and its arguments>
#   e.g. bash tools/sq_counters.sh r2_quad12 tools/bench_models.py f64 Quadrotor12 4096
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT="$PWD/gpurun_out/sq_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
P3="SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
# (round 4) the matrix pipe and the fp64 instruction mix: SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512 flops each (one
# v_mfma_f64_4x4x4_4b = 1, one v_mfma_f64_16x16x4 = 4: 16 clocks of the SIMD's fp64 pipe per unit)
P4="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/log$i.txt" 2>&1
done
python3 tools/sq_summary.py "$TAG" "$OUT"
cp "profiles/${TAG}_sq_counters.json" "$OUT/" 2>/dev/null
find "$OUT" -name "*.csv" -size +3M -delete
