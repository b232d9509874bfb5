set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_xr
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/a -- python3 tools/iter_modes.py 0 > /dev/null 2> $OUT/a.log || { tail -5 $OUT/a.log; exit 1; }
for k in k_xruns k_wvt_chain4 k_wvt_exact_w; do echo $k; python3 tools/pmc_per_dispatch.py $OUT/a $k | tail -1; done
