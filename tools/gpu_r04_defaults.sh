#!/bin/bash
# round 4: the whole GPU suite, the configs table, rocprof kernel statistics of the reference's default plan sizes (micro-tile
# kernels and, for comparison, the 16-candidate kernels), the micro-tile A/B table
TAG=${TAG:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 400 python tools/ab_micro.py lstm mlp > $OUT/ab_micro.jsonl 2> $OUT/ab_micro.err; echo "ab rc=$?"; tail -2 $OUT/ab_micro.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_micro -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_micro.err); echo "rocprof micro rc=$?"
(cd /tmp && L2A_MICRO=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_tile16 -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_tile16.err); echo "rocprof tile16 rc=$?"
for d in prof_micro prof_tile16; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv && head -8 $f | cut -c1-220; done
find $OUT -name "*.db" -delete 2>/dev/null; rm -rf $OUT/prof_micro $OUT/prof_tile16
if [ "${CONFIGS:-1}" = "1" ]; then
timeout 900 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"
python - <<PY
import json
for l in open("$OUT/configs.jsonl"):
    r = json.loads(l)
    print(r["config"][:100], {k: r[k] for k in ("kernel_ms", "frac_fp32_peak", "ms_per_call", "step_ms", "env_steps_per_s") if k in r})
PY
fi
