#!/bin/bash
# round 3: memory type of the exchange granules (L2A_XBUF_MODE 0 = hipMalloc, 1 = fine-grained, 3 = uncached)
TAG=${TAG:-r03x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for round in 1 2; do
  for mode in 0 1 3; do
    L2A_XBUF_MODE=$mode timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed "s/^{/{\"xbuf_mode\": $mode, /" | tee -a $OUT/ab.jsonl
    L2A_XBUF_MODE=$mode timeout 300 python tools/ab_lstm.py 2>> $OUT/ab.err | sed "s/^{/{\"xbuf_mode\": $mode, /" | tee -a $OUT/ab.jsonl
  done
done
tail -3 $OUT/ab.err
