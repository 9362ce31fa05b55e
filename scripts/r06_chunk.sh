#!/bin/bash
# round 6, VERDICT r5 item 1a: the sample-chunk-major schedule on one box — correctness, then the bench step at several chunk sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r06_chunk.txt
: > $O
timeout 600 python tools/chunk_check.py >> $O 2>&1
for mb in 0 64 120 180 0 240; do
  echo "== VSX_CHUNK_MB=$mb" >> $O
  VSX_CHUNK_MB=$mb timeout 600 python tools/ab_step.py --tag chunk$mb --steps 10 --rounds 3 2>&1 | grep -E '^\{|Error|error|Traceback' >> $O
done
for w in FWD BWD; do
  echo "== VSX_CHUNK_MB=120 only $w off" >> $O
  env VSX_CHUNK_MB=120 VSX_CHUNK_$w=0 timeout 600 python tools/ab_step.py --tag chunk120_no$w --steps 10 --rounds 3 2>&1 | grep -E '^\{|Error|error|Traceback' >> $O
done
cat $O
