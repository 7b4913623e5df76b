#!/bin/bash
# Timing-only ablation of the fused wide ResStack layer (resblock_act.hip): builds libvfx variants with -DVFX_RBA_ABL=mask
# HERE (hipcc cross-compiles), then on the GPU box: bash scripts/abl_resblock_act.sh run  ->  one line per variant.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/voicefixer_main_amd/csrc
MASKS="0 1 2 4 8 16 32 3 7"
if [ "${1:-build}" = build ]; then
  mkdir -p $ROOT/voicefixer_main_amd/abl
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -DVFX_RBA_ABL=$m -c $CS/resblock_act.hip -o /tmp/rba_$m.o || exit 1
    objs=$(ls $CS/build/*.o | grep -v resblock_act.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/voicefixer_main_amd/abl/libvfx_rba_$m.so $objs /tmp/rba_$m.o || exit 1
  done
  ls -la $ROOT/voicefixer_main_amd/abl
else
  for m in $MASKS; do
    VFX_LIB_PATH=$ROOT/voicefixer_main_amd/abl/libvfx_rba_$m.so python $ROOT/scripts/abl_resblock_act.py $m
  done
fi
