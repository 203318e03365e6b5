#!/bin/bash
# A/B timing of library builds on one GPU box, alternating runs so that box-to-box and thermal drift
# cancel: tools/ab_bench.sh <out-subdir> <rounds> libA.so libB.so ...   (paths relative to the repo root)
# Extra bench.py flags via AB_FLAGS.
set -u
OUT=gpurun_out/$1; ROUNDS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    FLACENC_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary ${AB_FLAGS:-} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_stats']; print('$lib', 'min', k['min'], 'median', k['median'], 'ms_per_step', d['ms_per_step'])" | tee -a $OUT/ab.txt
  done
done
