#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprof kernel trace.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01'
# Everything lands under gpurun_out/<tag>/ (scratch; copy what should be judged into profiles/).
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/host_cpu.txt
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -5 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
# keep the merged-back payload small
find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null
du -sh $OUT
