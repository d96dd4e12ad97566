#!/bin/bash
OUT=gpurun_out/${1:-r02k}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
# 1. LSTM early-sweep A/B (same box, alternating)
for round in 1 2 3; do
  for v in base noearly; do
    if [ "$v" = "base" ]; then unset L2A_LIB_PATH; else export L2A_LIB_PATH=$GRAFT_REPO_ROOT/learning_to_adapt_amd/libl2a_hip_$v.so; fi
    timeout 300 python tools/ab_lstm.py 2>> $OUT/ab.err | tee -a $OUT/ab_lstm.jsonl
  done
done
unset L2A_LIB_PATH
timeout 120 python tools/timeline_lstm.py 256 2000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_lstm.txt
# 2. draw-ahead wake-up A/B (same process, alternating)
timeout 300 python tools/probe_e2e.py --ab _defer_wake 2>> $OUT/ab.err | tee -a $OUT/ab_defer_wake.jsonl
# 3. the whole GPU suite
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -5 | tee $OUT/pytest_gpu_tail.txt
tail -3 $OUT/ab.err
