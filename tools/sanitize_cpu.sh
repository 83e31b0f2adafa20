#!/bin/bash
# tools/sanitize_cpu.sh -- the CPU suite under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers
# on the CPU build; GPU ASan is not available on the pool and must never be tried there).
#
# Builds oracle/ (test infrastructure) and jello_amd/host/ (the C++ mirror of Jello's host layers) with
# -fsanitize=address,undefined -fno-sanitize-recover=all into a scratch directory -- the product libraries in the tree are
# not touched -- and runs `pytest -m "not gpu"` with the sanitizer runtime preloaded into python.  Any finding aborts the
# test that triggered it.  The one test that calls jh_create (tests/test_abi.py: "no GPU -> loud failure") is deselected:
# it opens the HIP runtime, whose own allocations are not ASan-clean and not ours to judge.
#
#   tools/sanitize_cpu.sh            # whole CPU suite
#   tools/sanitize_cpu.sh -k image   # extra arguments go to pytest
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${SAN_OUT:-/tmp/jello_sanitize}"
mkdir -p "$OUT"
CXX="${CXX:-g++}"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g -O1"

# the product libraries must exist (the host library links against libjello_hip.so; nothing in it runs without a GPU)
[ -s "$ROOT/jello_amd/libjello_hip.so" ] || make -s -C "$ROOT/jello_amd/csrc" -j8  # (needs hipcc: skipped when the library is there -- CPU-only boxes)
FMAFLAG=$(grep -q -w fma /proc/cpuinfo 2>/dev/null && echo -mfma || true)
echo "[sanitize] building oracle -> $OUT/liboracle.so"
$CXX $SAN $FMAFLAG -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-strict-aliasing -fopenmp -shared \
    -o "$OUT/liboracle.so" "$ROOT/oracle/oracle.cpp"
echo "[sanitize] building host -> $OUT/libjello_host.so"
$CXX $SAN -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -I"$ROOT/include" -I"$ROOT/jello_amd/host" -shared \
    -o "$OUT/libjello_host.so" "$ROOT"/jello_amd/host/{encoding,scene,estimate,renderer,hip_engine,capi}.cpp \
    -L"$ROOT/jello_amd" -ljello_hip -Wl,-rpath,"$ROOT/jello_amd"

LIBASAN="$($CXX -print-file-name=libasan.so)"
LIBUBSAN="$($CXX -print-file-name=libubsan.so)"
cd "$ROOT"
# detect_leaks=0: python itself never frees its arenas.  OMP threads + ASan are fine; keep the pools small.
env LD_PRELOAD="$LIBASAN:$LIBUBSAN" \
    ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1" \
    UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1" \
    JELLO_HOST_LIB="$OUT/libjello_host.so" JELLO_ORACLE_LIB="$OUT/liboracle.so" OMP_NUM_THREADS=4 \
    python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider \
        --deselect tests/test_abi.py::test_no_gpu_means_loud_failure "$@"
echo "[sanitize] CPU suite clean under ASan + UBSan"
