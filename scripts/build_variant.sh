#!/bin/bash
# A variant of libvfx.so with extra -D flags on ONE OR MORE source files (comma-separated), for same-box A/B runs through VFX_LIB_PATH:
#   scripts/build_variant.sh <name> <file.hip[,file2.hip...]> <flags...>   ->  voicefixer_main_amd/abl/libvfx_<name>.so
# Built HERE (hipcc cross-compiles); voicefixer_main_amd/abl/ travels to the GPU box but stays out of the history.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/voicefixer_main_amd/csrc
name=$1; srcs=${2//,/ }; shift 2
make -C $CS -j8 > /dev/null
mkdir -p $ROOT/voicefixer_main_amd/abl /tmp/vfx_variant_$name
rm -f /tmp/vfx_variant_$name/*.o
skip="/ops_debug.o\|/investigate.o"
for src in $srcs; do
  base=${src%.*}
  x=""; [ "${src##*.}" = "cpp" ] && x="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include "$@" $x -c $CS/$src -o /tmp/vfx_variant_$name/$base.o &
  skip="$skip\|/$base.o"
done
wait
objs=$(ls $CS/build/*.o | grep -v "$skip")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libvfx.so -o $ROOT/voicefixer_main_amd/abl/libvfx_$name.so $objs /tmp/vfx_variant_$name/*.o
ls -la $ROOT/voicefixer_main_amd/abl/libvfx_$name.so
