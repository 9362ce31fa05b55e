#!/bin/bash
# usage: scripts/ktrace.sh <tag> <python args...>  → per-dispatch kernel durations (no PMC), summary printed
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$TAG -- python "$@" > $R/gpurun_out/kt_$TAG.log 2>&1
echo "rc=$?"
f=$(find $R/gpurun_out/kt_$TAG -name "*kernel_trace.csv" | head -1)
python $R/tools/ktrace.py "$f" ${KT_SORT:+--sort} | tail -${KT_TAIL:-80}
