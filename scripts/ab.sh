#!/bin/bash
# same-box A/B of library variants on the bench step: bash scripts/ab.sh "<variant> <variant> ..." [rounds] [extra ab_step args]
# ("base" = the shipped libvsx.so); results -> gpurun_out/ab.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
VARS="$1"; R=${2:-2}; shift; shift
: > gpurun_out/ab.txt
for r in $(seq 1 $R); do
  for v in $VARS; do
    lib=viscy_amd/libvsx_$v.so; [ "$v" == "base" ] && lib=viscy_amd/libvsx.so
    VSX_LIB=$PWD/$lib timeout 600 python tools/ab_step.py --tag $v "$@" 2>&1 | grep -E '^\{|Error|error' >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt
