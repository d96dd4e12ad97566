#!/bin/bash
# round 3: LSTM rollout kernel - workgroup lifetimes by XCD for the split (n=2000) and the chip-filling (n=4096) plan
TAG=${TAG:-r03v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for n in 2000 4096 2048; do
  timeout 200 python tools/timeline_lstm.py 256 $n > $OUT/timeline_lstm_$n.txt 2>&1; echo "timeline lstm $n rc=$?"
  grep -v amdgpu $OUT/timeline_lstm_$n.txt | cut -c1-330
done
