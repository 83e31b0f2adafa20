#!/bin/bash
# The sequential fall-back of k_flatten_items (taken when a batch overflows its LDS stack / piece list or a tree is
# deeper than FLQ_MAX_LEVEL) is never reached by the test scenes with the product capacities.  This builds the library
# with tiny capacities, so that most batches bail out, and runs the parity suite and a fuzz soak against the oracle.
# Third build: k_flatten_bbox with a grid of two workgroups, so that every wave strides over many line ranges (the
# product grid only does that beyond 16.7 M lines).  Fourth build: the temporary always in eight regions and every wave
# starting in region 0, so that regions fill up, allocations move on to the next one and the slots at a region's end stay empty
# (the product only gets there when a frame comes close to its line buffer's capacity); its soak also runs with line buffers of
# exactly the frame's size.
# (run on the GPU box; the product library is rebuilt with the default flags on ANY exit)
cd "$(dirname "$0")/.."
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for X in "-DFLQ_STACK=96u -DFLQ_LEAVES=80u" "-DFLQ_MAX_LEVEL=2u" "-DFB_MAX_BLOCKS=2u" "-DFL_SOAK_HOME0"; do
  make -s -C jello_amd/csrc EXTRA="$X" > /dev/null 2>&1 || exit 1
  echo "[$X]"
  timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
  timeout -k 10 300 python3 tools/parity_soak.py 100 300 2>&1 | tail -1
  [ "$X" = "-DFL_SOAK_HOME0" ] && TIGHT_LINES=1 timeout -k 10 300 python3 tools/parity_soak.py 100 200 2>&1 | tail -1
done
