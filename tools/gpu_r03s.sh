#!/bin/bash
# round 3: A/B of the XCD-exact group placement (L2A_XCD_ALIGN=1 default vs 0) + parity subset
TAG=${TAG:-r03s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== kernel A/B"
for round in 1 2 3; do
  L2A_XCD_ALIGN=0 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"xcd_align": 0, /' | tee -a $OUT/ab.jsonl
  L2A_XCD_ALIGN=1 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/^{/{"xcd_align": 1, /' | tee -a $OUT/ab.jsonl
done
tail -3 $OUT/ab.err
echo "== parity subset"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -rx -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log
echo "== bench"
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
