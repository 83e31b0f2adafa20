cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak5
PIDS=()
for i in 0 1 2 3; do
  timeout -k 10 900 python3 tools/parity_soak.py $((820000 + i * 1000)) 1000 > gpurun_out/soak5/fuzz_$i.log 2>&1 &
  PIDS+=($!)
done
RC=0
for P in "${PIDS[@]}"; do wait "$P" || RC=1; done
for i in 0 1 2 3; do tail -1 gpurun_out/soak5/fuzz_$i.log; done
[ $RC = 0 ] || exit 1
PIDS=()
for i in 0 1 2 3; do
  TIGHT_LINES=1 timeout -k 10 900 python3 tools/parity_soak.py $((830000 + i * 500)) 500 > gpurun_out/soak5/tight_$i.log 2>&1 &
  PIDS+=($!)
done
for P in "${PIDS[@]}"; do wait "$P" || RC=1; done
for i in 0 1 2 3; do tail -1 gpurun_out/soak5/tight_$i.log; done
[ $RC = 0 ] || exit 1
timeout -k 10 600 python3 tools/determinism.py > gpurun_out/soak5/determinism.log 2>&1; tail -6 gpurun_out/soak5/determinism.log
