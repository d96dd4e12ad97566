#!/bin/bash
# round 3, kernel iteration session: parity subset, A/B of variants (args: variant names), timelines of c2 and c3b
TAG=${TAG:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
echo "== parity subset"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log
echo "== kernel A/B"
for round in 1 2; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  for v in "$@"; do
    L2A_LIB_PATH=$LIBD/libl2a_hip_$v.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  done
done
tail -3 $OUT/ab.err
echo "== timelines"
timeout 120 python tools/timeline.py > $OUT/timeline_c2.txt 2>&1; echo "timeline c2 rc=$?"; grep -A3 "per-step" $OUT/timeline_c2.txt; grep "wave 0" $OUT/timeline_c2.txt | head -6
timeout 120 python tools/timeline.py c3b_ant_rs_n500_h10_pb5_3x512 > $OUT/timeline_c3b.txt 2>&1; echo "timeline c3b rc=$?"; cat $OUT/timeline_c3b.txt
du -sh $OUT
