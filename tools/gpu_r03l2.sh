#!/bin/bash
# experiment: both workgroups of a pair on one XCD (L2A_XCD_INTERLEAVE=2), records through the memory side (product
# library, sc1) or through that XCD's L2 (libl2a_hip_xl2.so: sc0 stores / loads)
TAG=${TAG:-r03l2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
echo "== parity of the sc0 variant on same-XCD pairs"
L2A_XCD_INTERLEAVE=2 L2A_LIB_PATH=$LIBD/libl2a_hip_xl2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "split or batching or golden or stale" --deselect "tests/test_gpu_parity.py::test_xxx" > $OUT/pytest_xl2.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_xl2.log
for round in 1 2; do
  L2A_XCD_INTERLEAVE=0 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"pairs": "XCD k | k+4, sc1", /' | tee -a $OUT/ab.jsonl
  L2A_XCD_INTERLEAVE=2 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"pairs": "same XCD, sc1", /' | tee -a $OUT/ab.jsonl
  L2A_XCD_INTERLEAVE=2 L2A_LIB_PATH=$LIBD/libl2a_hip_xl2.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"pairs": "same XCD, sc0 (L2)", /' | tee -a $OUT/ab.jsonl
done
tail -3 $OUT/ab.err
