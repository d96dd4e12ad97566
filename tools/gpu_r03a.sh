#!/bin/bash
# round 3, first GPU session: the batched rollout kernel - correctness, A/B against the round-2 library, timeline,
# parity report (fp32 and float64 return accumulation), short bench.
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
echo "== pytest -m gpu (no -x: collect every failure)"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== kernel A/B: round-2 library | batched (product) | product with one set at a time"
for round in 1 2; do
  L2A_LIB_PATH=$LIBD/libl2a_hip_r2base.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  L2A_BATCH=1 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/libl2a_hip.so/libl2a_hip.so batch=1/' | tee -a $OUT/ab.jsonl
  L2A_BATCH=2 timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | sed 's/libl2a_hip.so/libl2a_hip.so batch=2/' | tee -a $OUT/ab.jsonl
done
tail -3 $OUT/ab.err
echo "== timeline"
timeout 120 python tools/timeline.py > $OUT/timeline.txt 2>&1; echo "timeline rc=$?"; cat $OUT/timeline.txt
echo "== parity report"
timeout 600 python tools/parity_report.py > $OUT/parity_report.txt 2> $OUT/parity.err; echo "parity rc=$?"
grep -A60 "CEM case" $OUT/parity_report.txt | head -70
echo "== parity report, float64 return accumulation (experiment build)"
L2A_LIB_PATH=$LIBD/libl2a_hip_ret64.so timeout 600 python tools/parity_report.py > $OUT/parity_report_ret64.txt 2> $OUT/parity64.err; echo "parity64 rc=$?"
grep -A60 "CEM case" $OUT/parity_report_ret64.txt | grep "c5_\|case" | head -30
echo "== bench"
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
du -sh $OUT
