#!/bin/bash
# tools/asm_variant.sh MAXP VARIANT OCC [extra -D flags]: device assembly of one wave-kernel instance
# into /tmp/asm/w<MAXP>_<VARIANT>_occ<OCC>.s plus its register / scratch figures.
set -e
MP=$1; ST=$2; OCC=$3; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/asm
OUT=/tmp/asm/w${MP}_${ST}_occ${OCC}.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include \
  -I$ROOT/flacenc_rs_amd/csrc -DFLACENC_MAXP=$MP -DFLACENC_STEREO=$ST -DFLACENC_WAVE_OCC=$OCC "$@" --cuda-device-only \
  -S $ROOT/flacenc_rs_amd/csrc/qlpc_wave_inst.hip -o $OUT 2>&1 | grep -v "warning" || true
echo "$OUT: $(grep -E '^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_count|sgpr_spill_count)' $OUT | tr -s ' \n' ' ')"
