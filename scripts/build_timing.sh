#!/bin/bash
# Timing build of libvfx (-DVFX_TIMING: s_memtime stamps at the phase boundaries of the fused ResStack kernels (4-wave C = 128 / 256, persistent C = 64), written to the
# buffer named by VFX_TIMING_PTR): voicefixer_main_amd/abl/libvfx_timing.so.  Built HERE (hipcc cross-compiles); used on the
# GPU box by scripts/phase_timing.py.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/voicefixer_main_amd/csrc
make -C $CS -j8 > /dev/null
mkdir -p $ROOT/voicefixer_main_amd/abl /tmp/vfx_timing
for f in resblock resblock_w64 resblock_r128 resblock_rw block2d32; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -DVFX_TIMING -c $CS/$f.hip -o /tmp/vfx_timing/$f.o
done
objs=$(ls $CS/build/*.o | grep -v "/ops_debug.o\|/investigate.o\|/resblock.o\|/resblock_w64.o\|/resblock_r128.o\|/resblock_rw.o\|/block2d32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libvfx.so -o $ROOT/voicefixer_main_amd/abl/libvfx_timing.so $objs /tmp/vfx_timing/*.o
ls -la $ROOT/voicefixer_main_amd/abl/libvfx_timing.so
