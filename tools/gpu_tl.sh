#!/bin/bash
OUT=gpurun_out/${1:-tl}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for n in 2000 4096; do timeout 120 python tools/timeline_lstm.py 256 $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_lstm.txt; done
