#!/bin/bash
# round 4: micro-tile kernels - bit identity + timings (A/B by L2A_MICRO)
TAG=${TAG:-r04m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_rnn.py -m gpu -q -x --timeout 300 -k "micro" > $OUT/pytest_micro.log 2>&1; echo "pytest micro rc=$?"; tail -15 $OUT/pytest_micro.log
timeout 300 python tools/ab_micro.py > $OUT/ab_micro.jsonl 2> $OUT/ab_micro.err; echo "ab rc=$?"; cat $OUT/ab_micro.jsonl; tail -3 $OUT/ab_micro.err
