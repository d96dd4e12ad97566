#!/bin/bash
# horizon probes of one case across library variants: gpu_probe.sh <case> <variants...>
CASE=$1; shift
OUT=gpurun_out/${TAG:-probe}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
for round in 1 2; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/probe_horizon.py $CASE 5 10 20 40 2>> $OUT/err | tee -a $OUT/horizon.jsonl
  timeout 300 python tools/probe_horizon.py $CASE 5 10 20 40 2>> $OUT/err | tee -a $OUT/horizon.jsonl
  for v in "$@"; do
    L2A_LIB_PATH=$LIBD/libl2a_hip_$v.so timeout 300 python tools/probe_horizon.py $CASE 5 10 20 40 2>> $OUT/err | tee -a $OUT/horizon.jsonl
  done
done
