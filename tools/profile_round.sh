#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of the default bench command.
# Usage: bash tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
BP="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/trace $OUT/pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.log || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BP > /dev/null 2> $OUT/pmc_sq.log || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BP > /dev/null 2> $OUT/pmc_fetch.log || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- $BP > /dev/null 2> $OUT/pmc_write.log || exit 1
python3 tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
echo profiled
