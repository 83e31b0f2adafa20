#!/bin/bash
# Where k_flatten_items' instructions go: differential builds (VARIANT libraries, never the product one) without phase B
# (pieces) and without phases A + B (subdivision), one PMC pass each.  Results of the variants are WRONG by construction;
# only the counters of k_flatten_items are read.   usage (GPU box): tools/flatten_split.sh
cd "$(dirname "$0")/.."
R=$PWD
for v in product nob noab; do
  case $v in
    product) unset JELLO_HIP_LIB ;;
    nob) make -s -C jello_amd/csrc -j8 VARIANT=flnob EXTRA=-DFL_SPLIT_NO_B >/dev/null && export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_flnob.so ;;
    noab) make -s -C jello_amd/csrc -j8 VARIANT=flnoab EXTRA="-DFL_SPLIT_NO_A -DFL_SPLIT_NO_B" >/dev/null && export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_flnoab.so ;;
  esac
  rm -rf $R/gpurun_out/flsplit_$v; mkdir -p $R/gpurun_out/flsplit_$v
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/flsplit_$v -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph > $R/gpurun_out/flsplit_$v.log 2>&1)
  echo "== $v"; python3 profiles/pmc.py $R/gpurun_out/flsplit_$v/*/*counter_collection.csv --k=k_flatten_items,k_flatten_lines
done
