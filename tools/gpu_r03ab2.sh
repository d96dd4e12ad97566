#!/bin/bash
# A/B of the recurrent kernels: product library against libl2a_hip_prev.so + the recurrent parity tests
TAG=${TAG:-r03ab2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
timeout 900 python -m pytest tests/test_rnn.py tests/test_gpu_random_shapes.py tests/test_checkpoint.py -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for round in 1 2 3; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_prev.so timeout 300 python tools/ab_lstm.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  timeout 300 python tools/ab_lstm.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
done
