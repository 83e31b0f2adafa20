#!/bin/bash
# Where k_flatten_lines' time goes: differential builds (VARIANT libraries, never the product one) without the Euler evaluation
# (FL_LSPLIT=1) and without the stores (FL_LSPLIT=2), kernel time and one PMC pass each.  Results of the variants are WRONG by
# construction; only the kernel's duration and counters are read.   usage (GPU box): tools/lines_split.sh
cd "$(dirname "$0")/.."
R=$PWD
for v in product lsplit1 lsplit2; do
  case $v in
    product) unset JELLO_HIP_LIB ;;
    *) make -s -C jello_amd/csrc -j8 VARIANT=$v EXTRA=-DFL_LSPLIT=${v#lsplit} >/dev/null && export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_$v.so ;;
  esac
  rm -rf $R/gpurun_out/lsplit_$v; mkdir -p $R/gpurun_out/lsplit_$v
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lsplit_$v/t -- python3 $R/bench.py --steps 10 --warmup 2 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph --in-flight 1 > $R/gpurun_out/lsplit_$v.log 2>&1)
  echo "== $v"; python3 profiles/kstats.py $(ls $R/gpurun_out/lsplit_$v/t/*/*kernel_stats.csv | head -1) 60 | grep -E "k_flatten"
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/lsplit_$v/p -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph --in-flight 1 >> $R/gpurun_out/lsplit_$v.log 2>&1)
  python3 profiles/pmc.py $R/gpurun_out/lsplit_$v/p/*/*counter_collection.csv --k=k_flatten_lines
  rm -rf $R/gpurun_out/lsplit_$v
done
