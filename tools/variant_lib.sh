#!/bin/bash
# tools/variant_lib.sh NAME SRC.cpp [extra -D flags]: an A/B library ab/libflacenc_hip_NAME.so that
# differs from the current build only in one translation unit compiled with extra flags (the other objects
# come from flacenc_rs_amd/csrc/build; run `make` there first).
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/flacenc_rs_amd/csrc
mkdir -p $ROOT/ab /tmp/variant_$NAME
BASE=$(basename $SRC .cpp)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include -I$C \
  "$@" -x hip -c $C/$SRC -o /tmp/variant_$NAME/$BASE.o 2>&1 | grep -v warning || true
OBJS="$(ls $C/build/*.o | grep -v "/$BASE.o" | grep -v "/flacenc_hip_api.o") $C/build/hooks/flacenc_hip_api.o"  # (A/B libraries carry the debug hooks: the tools that load them use both)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME/$BASE.o -o $ROOT/ab/libflacenc_hip_$NAME.so
echo ab/libflacenc_hip_$NAME.so
