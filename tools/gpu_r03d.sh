#!/bin/bash
# round 3, session 4: full GPU suite with the O4 output path, A/B, horizon probes (per-launch vs per-step cost)
TAG=${TAG:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
echo "== kernel A/B"
for round in 1 2 3; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  for v in "$@"; do
    L2A_LIB_PATH=$LIBD/libl2a_hip_$v.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  done
done
tail -3 $OUT/ab.err
echo "== horizon probes"
for c in c3b_ant_rs_n500_h10_pb5_3x512 c1_hc_rs_n500_h10_e1 c2_hc_rs_n2000_h30_e5; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/probe_horizon.py $c 2>> $OUT/ab.err | tee -a $OUT/horizon.jsonl
  timeout 300 python tools/probe_horizon.py $c 2>> $OUT/ab.err | tee -a $OUT/horizon.jsonl
done
echo "== timeline c2"
timeout 120 python tools/timeline.py > $OUT/timeline_c2.txt 2>&1; echo "timeline c2 rc=$?"; grep -A5 "per-step" $OUT/timeline_c2.txt | cut -c1-1300; grep "wave 0" $OUT/timeline_c2.txt
du -sh $OUT
