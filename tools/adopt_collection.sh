#!/bin/bash
# Copy what tools/collect_round.sh <tag> left under gpurun_out/ into profiles/ under the round's names:
#   tools/adopt_collection.sh r03j r03 [bench.json to use instead of the collection's own]
set -e
TAG=$1; R=$2; BENCH=${3:-gpurun_out/collect_$TAG/bench.json}
P=gpurun_out/profile_$TAG; S=gpurun_out/stages_$TAG; C=gpurun_out/collect_$TAG
cp $BENCH profiles/${R}_bench.json
cp $P/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $P/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $P/kernel_stats_timed.csv profiles/${R}_kernel_stats_timed.csv
cp $P/pmc_summary.json profiles/${R}_pmc_summary.json
cp $P/headline_pmc.json profiles/headline_pmc.json
[ -s $C/headline_phases.json ] && cp $C/headline_phases.json profiles/headline_phases.json
cp $S/bench_pack.json profiles/${R}_stages_bench_pack.json
cp $S/kernel_stats.csv profiles/${R}_stages_kernel_stats.csv
cp $C/configs.json profiles/${R}_configs.json
cp $C/bigblock_pmc_n8192_p24.txt profiles/${R}_config3_n8192_p24_pmc.txt
cp $C/bigblock_pmc_n8192_p32.txt profiles/${R}_config3_n8192_p32_pmc.txt
for d in n8192_p24 n8192_p32 n16384_p24 n16384_p32; do
  f=$(ls $C/bigblock/$d/runc/*kernel_stats.csv 2>/dev/null | head -1)
  cfg=$([ "${d:0:5}" = "n8192" ] && echo config3 || echo config5)
  [ -n "$f" ] && cp $f profiles/${R}_${cfg}_${d}_kernel_stats.csv
done
git status --short profiles | head -30
