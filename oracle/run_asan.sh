#!/bin/sh
# Runs the CPU test files that exercise the oracle and the harness generator against their
# AddressSanitizer + UndefinedBehaviorSanitizer builds (make -C oracle asan).  CPU only.
# The sanitizer runtime must be the first library of the process, hence LD_PRELOAD; leak checking
# is off because the Python interpreter itself never frees everything.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
ASAN_RT=$(gcc -print-file-name=libasan.so)
UBSAN_RT=$(gcc -print-file-name=libubsan.so)
cd "$ROOT"
env LD_PRELOAD="$ASAN_RT:$UBSAN_RT" \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    CUEMBED_ORACLE_LIB="$HERE/_asan/libcuembed_oracle_asan.so" \
    CUEMBED_HARNESS_LIB="$HERE/_asan/libcuembed_harness_asan.so" \
    python3 -m pytest tests/test_oracle_golden.py tests/test_harness_datagen.py -x -q -p no:cacheprovider "$@"
