#!/bin/bash
# round-2 first GPU session: parity tests, bench line, host RNG scaling on the box's CPUs
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/host_cpu.txt
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== bench (driver-style 20/5)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2>> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench_20_5.json
echo "== rng"
timeout 300 python tools/bench_rng.py > $OUT/rng.jsonl 2>&1; cat $OUT/rng.jsonl
echo "== configs"
timeout 600 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"; tail -30 $OUT/configs.jsonl; tail -5 $OUT/configs.err
