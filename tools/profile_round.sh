#!/bin/bash
# Round profile of the default bench workload: rocprofv3 kernel-trace stats, then the PMC passes (each its
# own run, --kernel-trace only), then the per-launch figures bench.py attaches to its roofline object.
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh r02
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# the timed region alone: the last 20 dispatches of each kernel (the --stats average above also holds the
# untimed clock spin-up and warm-up launches)
python3 $GRAFT_REPO_ROOT/tools/trace_timed_stats.py $OUT/trace 20 > $OUT/kernel_stats_timed.csv
bash $GRAFT_REPO_ROOT/tools/pmc_profile.sh profile_$TAG/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/make_headline_pmc.py $OUT/pmc_summary.json $TAG > $OUT/headline_pmc.json
cat $OUT/kernel_stats.csv
cat $OUT/kernel_stats_timed.csv
cat $OUT/headline_pmc.json
