#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call: tools/collect_round.sh r02
set -u
TAG=$1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$GRAFT_REPO_ROOT/gpurun_out/collect_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py > $O/bench.json 2> $O/bench.err
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
bash tools/profile_stages.sh $TAG > $O/profile_stages.log 2>&1
python3 tools/bench_configs.py > $O/configs.json 2> $O/configs.txt
tools/prof_bigblock.sh collect_$TAG/bigblock > $O/bigblock.txt 2>&1
tools/pmc_insts_bigblock.sh collect_$TAG/bigblock_pmc24 8192 24 > $O/bigblock_pmc_n8192_p24.txt 2>&1
tools/pmc_insts_bigblock.sh collect_$TAG/bigblock_pmc32 8192 32 > $O/bigblock_pmc_n8192_p32.txt 2>&1
# per-phase instruction counts of the headline kernel (needs ab/libflacenc_exit{0..3}.so from tools/build_exit_variants.sh, run
# in the build container before this call; phase_insts.sh refuses libraries of other sources)
bash tools/phase_insts.sh collect_$TAG/phases > $O/phases.txt 2>&1
python3 tools/make_headline_phases.py $O/phases/summary.txt $TAG > $O/headline_phases.json 2>> $O/phases.txt
tail -c 1500 $O/bench.json; cat $O/configs.txt; cat $O/bigblock.txt | tail -30
