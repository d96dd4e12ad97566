#!/bin/bash
OUT=gpurun_out/${1:-r02j}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_rnn.py -m gpu -q --timeout 300 -x 2>&1 | tail -3
for n in 2000 4096; do timeout 120 python tools/timeline_lstm.py 256 $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_lstm.txt; done
timeout 600 python tools/bench_configs.py 2> $OUT/configs.err | grep -i "rebal\|lstm" | grep -v valu | tee $OUT/configs_rnn.jsonl
