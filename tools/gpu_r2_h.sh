#!/bin/bash
TAG=${1:-r02h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== pytest rnn"
timeout 900 python -m pytest tests/test_rnn.py tests/test_checkpoint.py tests/test_closed_loop.py tests/test_gpu_random_shapes.py -m gpu -q --timeout 300 -x > $OUT/pytest_rnn.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_rnn.log
echo "== recurrent configs"
timeout 600 python tools/bench_configs.py 2> $OUT/configs.err | grep -i "rebal\|lstm" | tee $OUT/configs_rnn.jsonl
tail -3 $OUT/configs.err
