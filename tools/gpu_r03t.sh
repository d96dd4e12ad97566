#!/bin/bash
# round 3: XCD placement by weight-sharing unit: A/B (L2A_XCD_ALIGN 1/0), workgroup lifetimes by XCD, parity
TAG=${TAG:-r03t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== kernel A/B"
for round in 1 2; do
  L2A_XCD_ALIGN=0 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"xcd_align": 0, /' | tee -a $OUT/ab.jsonl
  L2A_XCD_ALIGN=1 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"xcd_align": 1, /' | tee -a $OUT/ab.jsonl
done
for al in 1 0; do
  L2A_XCD_ALIGN=$al timeout 120 python tools/timeline.py c3b_ant_rs_n500_h10_pb5_3x512 > $OUT/timeline_c3b_align$al.txt 2>&1; sed -n 1,14p $OUT/timeline_c3b_align$al.txt
done
L2A_XCD_ALIGN=1 timeout 120 python tools/timeline.py > $OUT/timeline_c2.txt 2>&1; sed -n 1,13p $OUT/timeline_c2.txt
L2A_XCD_ALIGN=1 timeout 120 python tools/timeline.py c1_hc_rs_n500_h10_e1 > $OUT/timeline_c1.txt 2>&1; sed -n 1,8p $OUT/timeline_c1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_shapes.py -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
