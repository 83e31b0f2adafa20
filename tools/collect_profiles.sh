#!/bin/bash
# Everything profiles/ holds for one state of the code (run on the GPU box): kernel statistics and bench lines of C3,
# C4 and the nested C4 variant, the fine counter summaries, PTCL statistics, the other-scene timings.
#   tools/collect_profiles.sh <commit-id>   ->  gpurun_out/collect/
R="$(cd "$(dirname "$0")/.." && pwd)"
COMMIT=${1:-unknown}
# PART=1 (counters, kernel statistics, bench lines) / PART=2 (fine split, PTCL statistics, other scenes, round-4 extras) / unset = both:
# a gpurun call is limited to 20 minutes
PART=${PART:-all}
O=$R/gpurun_out/collect
[ "$PART" != 2 ] && rm -rf "$O"
mkdir -p "$O"
cd "$R"
if [ "$PART" != 2 ]; then
# counters first: the bench lines below report roofline.traffic from profiles/fine_counters*.json only if those were measured on
# the kernel sources being run (here: on the box's copy of profiles/; copy gpurun_out/collect/fine_counters*.json home afterwards)
bash tools/pmc_fine.sh "$COMMIT" > "$O/pmc_c3.log" 2>&1 && echo "pmc c3 done"
bash tools/pmc_fine.sh "$COMMIT" --scene c4 > "$O/pmc_c4.log" 2>&1 && echo "pmc c4 done"
bash tools/pmc_fine.sh "$COMMIT" --scene c4n > "$O/pmc_c4n.log" 2>&1 && echo "pmc c4n done"
cp gpurun_out/fine_counters*.json profiles/ 2>/dev/null
for S in c3 c4 c4n; do
  bash tools/kprof.sh col_$S --scene $S > "$O/${S}_summary.txt" 2>&1 || exit 1
  cp gpurun_out/kprof_col_$S/kernel_stats.csv "$O/${S}_kernel_stats.csv"
  timeout -k 10 300 python3 bench.py --scene $S > "$O/${S}_bench.json" 2> "$O/${S}_bench.err" || exit 1
  echo "bench $S done"
done
timeout -k 10 300 python3 bench.py --aa msaa8 --no-cpu-baseline > "$O/c3_msaa8_bench.json" 2>/dev/null
timeout -k 10 300 python3 bench.py --aa msaa16 --no-cpu-baseline > "$O/c3_msaa16_bench.json" 2>/dev/null
fi
[ "$PART" = 1 ] && exit 0
bash tools/fine_split.sh "$COMMIT" > "$O/fine_split.log" 2>&1 && cp gpurun_out/fine_split.json "$O/" && echo "fine split done"
cp gpurun_out/fine_counters*.json "$O/" 2>/dev/null
for S in c3 c4 c4n; do timeout -k 10 300 python3 tools/ptcl_stats.py $S > "$O/ptcl_stats_$S.json" 2>/dev/null; done
echo "ptcl stats done"
( timeout -k 10 300 python3 tools/time_configs.py; timeout -k 10 300 python3 tools/time_shapes.py ) > "$O/other_scenes.txt" 2>&1
echo "other scenes done"
# round 4: the small configurations as bench lines, the counters of the flatten and tile-stage kernels, the split of k_flatten_items
for S in c1 c2; do timeout -k 10 300 python3 bench.py --scene $S --no-cpu-baseline > "$O/${S}_bench.json" 2>/dev/null; done
bash tools/pmc_flatten.sh > "$O/pmc_flatten_tile_kernels.txt" 2>&1 && echo "pmc flatten / tile kernels done"
bash tools/flatten_split.sh > "$O/flatten_split.txt" 2>&1 && echo "flatten split done"
# round 5: k_flatten_lines split, frames in flight, and -- if the round-5 library was built next to the product one
# (jello_amd/libjello_hip_r05.so: `git archive 5df0f7a jello_amd/csrc include | tar -x -C /tmp/r05src && make -C /tmp/r05src/jello_amd/csrc`,
# copied in) -- the same-box A/B of every kernel against it
bash tools/lines_split.sh > "$O/lines_split.txt" 2>&1 && echo "lines split done"
for S in c3 c4 c4n; do timeout -k 10 250 python3 tools/frames_in_flight.py --scene $S --max-in-flight 3 > "$O/frames_in_flight_$S.json" 2>/dev/null; done
echo "frames in flight done"
if [ -s jello_amd/libjello_hip_r05.so ]; then
  ( bash tools/ab_kernels.sh "k_" r05 product; AB_ARGS="--scene c4" bash tools/ab_kernels.sh "k_" r05 product; AB_ARGS="--scene c4n" bash tools/ab_kernels.sh "k_fine|k_coarse|k_clip" r05 product; AB_ARGS="--aa msaa8" bash tools/ab_kernels.sh "k_fine" r05 product; AB_ARGS="--aa msaa16" bash tools/ab_kernels.sh "k_fine" r05 product ) > "$O/ab_r05_vs_r06.txt" 2>&1
  echo "A/B against round 5 done"
fi
