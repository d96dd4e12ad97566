#!/bin/bash
# round 3, kernel iteration session 3: probes, A/B of the exchange / prefetch variants, c3b timeline old vs new
TAG=${TAG:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
echo "== probes"; tools/probes/permlane_sum | tail -3
echo "== parity subset (product)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for v in "$@"; do
  echo "== split / batch bit-identity with variant $v"
  L2A_LIB_PATH=$LIBD/libl2a_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "bit_identical or golden or stale or degrad" 2>&1 | tail -2
done
echo "== kernel A/B"
for round in 1 2 3; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  for v in "$@"; do
    L2A_LIB_PATH=$LIBD/libl2a_hip_$v.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  done
done
tail -3 $OUT/ab.err
echo "== timelines"
timeout 120 python tools/timeline.py > $OUT/timeline_c2.txt 2>&1; echo "timeline c2 rc=$?"; grep -A5 "per-step" $OUT/timeline_c2.txt | cut -c1-1200
timeout 120 python tools/timeline.py c3b_ant_rs_n500_h10_pb5_3x512 > $OUT/timeline_c3b.txt 2>&1; echo "timeline c3b rc=$?"; grep -v amdgpu.ids $OUT/timeline_c3b.txt
L2A_TIMELINE_LIB=$LIBD/libl2a_hip_r2timeline.so timeout 120 python tools/timeline.py c3b_ant_rs_n500_h10_pb5_3x512 > $OUT/timeline_c3b_r2.txt 2>&1; echo "timeline c3b (round-2 library) rc=$?"; grep -v amdgpu.ids $OUT/timeline_c3b_r2.txt
du -sh $OUT
