#!/bin/bash
# tools/variant_bigres.sh NAME K NLB [extra -D flags]: an A/B library ab/libflacenc_hip_NAME.so that
# differs from the current build in one bigblock_residual_kernel instance compiled with extra flags
set -e
NAME=$1; K=$2; NLB=$3; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/flacenc_rs_amd/csrc
mkdir -p $ROOT/ab /tmp/variant_$NAME
BASE=qlpc_bigres_inst_${K}_${NLB}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include -I$C \
  -DFLACENC_BIG_K=$K -DFLACENC_BIG_NLB=$NLB "$@" -c $C/qlpc_bigblock_residual_inst.hip -o /tmp/variant_$NAME/$BASE.o 2>&1 | grep -v warning || true
OBJS="$(ls $C/build/*.o | grep -v "/$BASE.o" | grep -v "/flacenc_hip_api.o") $C/build/hooks/flacenc_hip_api.o"  # (A/B libraries carry the debug hooks: the tools that load them use both)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME/$BASE.o -o $ROOT/ab/libflacenc_hip_$NAME.so
echo ab/libflacenc_hip_$NAME.so
