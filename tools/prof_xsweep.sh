set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_x1
rm -rf $OUT; mkdir -p $OUT
for ab in 0 1 2; do
P="python3 tools/xsweep_run.py 2e6 $ab"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/a${ab}_sq -- $P > /dev/null 2> $OUT/a${ab}_sq.log || { tail -5 $OUT/a${ab}_sq.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/a${ab}_mem -- $P > /dev/null 2> $OUT/a${ab}_mem.log || { tail -5 $OUT/a${ab}_mem.log; exit 1; }
for d in sq mem; do python3 tools/pmc_per_dispatch.py $OUT/a${ab}_$d k_wvt_exact4 | tail -1 > $OUT/a${ab}_$d.txt; done
echo "ablate $ab"; cat $OUT/a${ab}_sq.txt $OUT/a${ab}_mem.txt
done
