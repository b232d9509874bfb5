#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of the default bench command, and the
# VALU issue-cost micro-benchmark.   Usage: bash tools/profile_round2.sh <tag>   -> gpurun_out/prof_<tag>/...
# Counters go in passes of <= 8 SQ slots; FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md).
# No PMC pass is combined with a hip/hsa/memory trace domain.  Summarise afterwards (in the repo, CPU only) with
#   python3 tools/roofline_valu.py gpurun_out/prof_<tag> --round N
set -o pipefail
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-relax"
BP="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-relax"
rm -rf $OUT/trace $OUT/pmc_*
pmc() {   # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BP > /dev/null 2> $OUT/pmc_$name.log || { echo "pass $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
  echo "pass $name done"
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.log || exit 1
echo "trace done"
pmc sq      SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc f64     SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU
pmc f32     SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
pmc lds     SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pmc grbm    GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pmc fetch   FETCH_SIZE
pmc write   WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
[ -x tools/ubench/valu_cost ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/ubench/valu_cost tools/ubench/valu_cost.hip || { echo "valu_cost build failed"; exit 1; }
./tools/ubench/valu_cost 4 > $OUT/valu_cost_w4.json 2> $OUT/valu_cost.log || { echo "valu_cost failed"; exit 1; }
./tools/ubench/valu_cost 8 > $OUT/valu_cost_w8.json 2>> $OUT/valu_cost.log
echo profiled
