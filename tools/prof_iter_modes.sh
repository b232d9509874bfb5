set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_im
rm -rf $OUT; mkdir -p $OUT
for md in 0 2 1; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/m$md -- python3 tools/iter_modes.py $md > /dev/null 2> $OUT/m$md.log || { tail -5 $OUT/m$md.log; exit 1; }
echo "mode $md"; python3 tools/pmc_per_dispatch.py $OUT/m$md k_iter | tail -1
done
