#!/bin/bash
# tools/soak_flatten_fast.py in NPROC processes over adjoining seed ranges (the soak is bound by building the fuzz scenes on the
# host, not by the GPU; at most 6 processes may use the card).  Run on the GPU box; builds the check variant if it is missing.
#   tools/soak_flatten_fast.sh <first_seed> <seeds_per_process> [nproc=4]   ->  gpurun_out/ffsoak_<first_seed>_<i>.log
R="$(cd "$(dirname "$0")/.." && pwd)"
FIRST=${1:?first seed}; PER=${2:?seeds per process}; NPROC=${3:-4}
[ "$NPROC" -le 5 ] || { echo "at most 5 processes"; exit 2; }
cd "$R"
LIB=$R/jello_amd/libjello_hip_ffcheck.so
mkdir -p gpurun_out
[ -s "$LIB" ] || make -C jello_amd/csrc VARIANT=ffcheck EXTRA=-DFL_FAST_CHECK > gpurun_out/ffcheck_build.log 2>&1 || exit 1
PIDS=()
for ((i = 0; i < NPROC; i++)); do
  JELLO_HIP_LIB=$LIB timeout -k 10 ${SOAK_TIMEOUT:-1000} python3 tools/soak_flatten_fast.py $((FIRST + i * PER)) "$PER" \
    > "gpurun_out/ffsoak_${FIRST}_$i.log" 2>&1 &
  PIDS+=($!)
done
RC=0
for P in "${PIDS[@]}"; do wait "$P" || RC=1; done
for ((i = 0; i < NPROC; i++)); do tail -1 "gpurun_out/ffsoak_${FIRST}_$i.log"; done
exit $RC
