#!/bin/bash
# CPU-side AddressSanitizer + UBSan run of the C host library and the oracle (GPU sanitizers are not
# available on the pool).  Builds sanitized copies in place, runs the CPU tests that load them through
# ctypes with the sanitizer runtimes preloaded, then restores the normal builds.
set -e
cd "$(dirname "$0")/.."
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
make -C toycluster_amd/host clean > /dev/null
# every artefact the test session looks for must exist (tests/conftest.py would otherwise start a build under the
# preloaded sanitizer runtimes): build the whole host directory sanitized, not only the library under test
make -C toycluster_amd/host CC="gcc $SAN" \
     CFLAGS="-std=c99 -O1 -g -Wall -fPIC -D_POSIX_C_SOURCE=200809L $SAN" > /dev/null
gcc -std=c99 -O1 -g -fopenmp -fPIC -fno-strict-aliasing $SAN -shared -o oracle/libtcoracle.so oracle/tc_oracle.c oracle/tc_oracle_sub.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$ASAN $UBSAN" OMP_NUM_THREADS=4 python -m pytest -q -s \
    tests/test_host_io.py tests/test_host_setup.py tests/test_reassign.py tests/test_substructure.py tests/test_oracle.py \
    2>&1 | grep -i "runtime error\|AddressSanitizer\|passed\|failed" || true
make -C toycluster_amd/host clean > /dev/null; make -C toycluster_amd/host > /dev/null
touch oracle/tc_oracle.c; make -C oracle > /dev/null
echo "normal builds restored"
