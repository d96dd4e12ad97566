#!/bin/bash
# round 3: robustness session - full GPU suite, random-shape sweep (150 seeds x MLP / LSTM), soak, recurrent A/B
TAG=${TAG:-r03j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
echo "== random shapes, 150 seeds"
L2A_RANDOM_SEEDS=150 timeout 1200 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q --timeout 600 > $OUT/random_shapes.log 2>&1; echo "random rc=$?"; tail -4 $OUT/random_shapes.log
echo "== soak 45 s"
timeout 300 python tools/soak.py 45 > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -6 $OUT/soak.txt
echo "== recurrent kernel shapes"
timeout 300 python tools/bench_configs.py 2> /dev/null | grep -i "rebal\|lstm" | tee $OUT/configs_rnn.jsonl | cut -c1-200
du -sh $OUT
