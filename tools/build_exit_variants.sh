#!/bin/bash
# The diagnostic builds tools/phase_insts.sh measures -- ab/libflacenc_exit{0..3}.so: the headline instance ending after
# phase k (-DFLACENC_EXIT_AFTER=k) -- built from the CURRENT sources (run it here, before the gpurun call), stamped with
# the kernel source fingerprint bench.py uses so that phase_insts.sh can refuse stale ones.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for k in 0 1 2 3; do
  bash tools/variant_wave.sh exit$k 8 2 -DFLACENC_EXIT_AFTER=$k -DFLACENC_HIP_DEBUG_HOOKS | tail -1
  mv ab/libflacenc_hip_exit$k.so ab/libflacenc_exit$k.so
done
python3 -c "import bench; print(bench.kernel_source_sha())" > ab/exit_variants.sha
echo "ab/exit_variants.sha $(cat ab/exit_variants.sha)"
