#!/bin/bash
# usage: gpu_ab.sh <tag> <variant names...>   (variants built by tools/build_variant.py; "base" = the product library)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for round in 1 2; do
for v in base "$@"; do
  if [ "$v" = "base" ]; then unset L2A_LIB_PATH; else export L2A_LIB_PATH=$GRAFT_REPO_ROOT/learning_to_adapt_amd/libl2a_hip_$v.so; fi
  timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
done
done
unset L2A_LIB_PATH
for v in "$@"; do
  echo "== parity subset with variant $v"
  L2A_LIB_PATH=$GRAFT_REPO_ROOT/learning_to_adapt_amd/libl2a_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 2>&1 | tail -3
done
tail -3 $OUT/ab.err
