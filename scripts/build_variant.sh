#!/bin/bash
# A variant of libvfx.so with extra -D flags on ONE source file, for same-box A/B runs through VFX_LIB_PATH:
#   scripts/build_variant.sh <name> <file.hip> <flags...>   ->  voicefixer_main_amd/abl/libvfx_<name>.so
# Built HERE (hipcc cross-compiles); voicefixer_main_amd/abl/ travels to the GPU box but stays out of the history.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/voicefixer_main_amd/csrc
name=$1; src=$2; shift 2
make -C $CS -j8 > /dev/null
mkdir -p $ROOT/voicefixer_main_amd/abl /tmp/vfx_variant_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include "$@" -c $CS/$src -o /tmp/vfx_variant_$name/$base.o
objs=$(ls $CS/build/*.o | grep -v "/ops_debug.o\|/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libvfx.so -o $ROOT/voicefixer_main_amd/abl/libvfx_$name.so $objs /tmp/vfx_variant_$name/$base.o
ls -la $ROOT/voicefixer_main_amd/abl/libvfx_$name.so
