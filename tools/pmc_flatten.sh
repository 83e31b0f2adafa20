#!/bin/bash
# Issue statistics of the flatten kernels (one PMC pass, no trace domains besides the kernel trace).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p $R/gpurun_out/flpmc
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/flpmc -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph > $R/gpurun_out/flpmc.log 2>&1)
python3 profiles/pmc.py $R/gpurun_out/flpmc/*/*counter_collection.csv --k=k_flatten_classify,k_flatten_items,k_flatten_lines,k_flatten_bbox,k_pc_count,k_pc_paths,k_pc_emit,k_pc_rank_small,k_coarse,k_path_tiling,k_scan_lookback
