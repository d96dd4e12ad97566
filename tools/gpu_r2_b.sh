#!/bin/bash
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
echo "== probe"
timeout 300 python tools/probe_e2e.py > $OUT/probe.jsonl 2> $OUT/probe.err; cat $OUT/probe.jsonl; tail -5 $OUT/probe.err
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== configs"
timeout 600 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"; grep -i "end to end\|GrBAL\|CEM" $OUT/configs.jsonl; tail -5 $OUT/configs.err
