#!/bin/bash
# Compile conv.hip to gfx950 assembly and print per-kernel register / scratch statistics.
cd /root/repo/voicefixer_main_amd/csrc || exit 1
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -S --cuda-device-only -o /tmp/asm/conv.s conv.hip 2>&1 | grep -v "warning: argument unused" | head -30
grep "^    .private_segment_fixed_size\|^    .name:.*k_conv\|^    .vgpr_count\|^    .vgpr_spill\|^    .sgpr_count" /tmp/asm/conv.s | paste - - - - - | awk '{print $2, "scratch", $4, "sgpr", $6, "vgpr", $8, "spill", $10}'
