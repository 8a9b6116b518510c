#!/bin/bash
# The CPU test suite over the ASan + UBSan builds of the host shim and the oracle (SURVEY section 5 asks for a sanitizer run).
# The sanitizer runtime must be loaded before python: LD_PRELOAD. Leak checking is off (CPython itself "leaks" at exit).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/oracle" -s asan
make -C "$ROOT/ml-ease_amd/host" -s asan
export MLX_ASAN=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cd "$ROOT" && python -m pytest tests -q -m "not gpu" "$@"
