#!/bin/bash
# rocprofv3 kernel-trace stats of the widened pipeline (fixed-LPC candidate + frame packing).
# usage (on the GPU box, via gpurun): bash tools/profile_stages.sh r01
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/stages_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_pack.py --use-fixed --frames 24576 > $OUT/bench_pack.json 2> $OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/kernel_stats.csv
tail -1 $OUT/bench_pack.json
