#!/bin/bash
# Round-6 soak on the GPU box: fuzz scenes (all three coverage modes, clips, gradients, images) through the HIP pipeline and the oracle in four
# processes, the same with the smallest line buffers the frames fit, then run-to-run determinism at full size (C3 area, C4 msaa8).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak6
PIDS=()
for i in 0 1 2 3; do
  timeout -k 10 1000 python3 tools/parity_soak.py $((${SOAK_BASE:-920000} + i * 1500)) 1500 > gpurun_out/soak6/fuzz_$i.log 2>&1 &
  PIDS+=($!)
done
RC=0
for P in "${PIDS[@]}"; do wait "$P" || RC=1; done
for i in 0 1 2 3; do tail -1 gpurun_out/soak6/fuzz_$i.log; done
[ $RC = 0 ] || exit 1
PIDS=()
for i in 0 1 2 3; do
  TIGHT_LINES=1 timeout -k 10 900 python3 tools/parity_soak.py $((${SOAK_BASE:-920000} + 10000 + i * 500)) 500 > gpurun_out/soak6/tight_$i.log 2>&1 &
  PIDS+=($!)
done
for P in "${PIDS[@]}"; do wait "$P" || RC=1; done
for i in 0 1 2 3; do tail -1 gpurun_out/soak6/tight_$i.log; done
[ $RC = 0 ] || exit 1
timeout -k 10 600 python3 tools/determinism.py > gpurun_out/soak6/determinism.log 2>&1; tail -6 gpurun_out/soak6/determinism.log
