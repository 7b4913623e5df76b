#!/bin/bash
# Sample GPU clock / power while a command runs:  scripts/smi_sample.sh <out.txt> <command...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|Power\|GPU use\|fclk\|mclk" | tr '\n' ' ' >> "$out"; echo >> "$out"; sleep 0.4; done ) &
sampler=$!
"$@"
kill $sampler 2>/dev/null
