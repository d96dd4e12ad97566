#!/bin/bash
# Mid-round GPU session: full GPU suite, bench, the controller-step probes (stage tables + 5000-call distribution), N = 2 code path.
TAG=${1:-mid}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
lscpu | grep -E "Model name|^CPU\(s\)" > $OUT/host_cpu.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20.json 2>> $OUT/bench.err; echo "bench20 rc=$?"
cut -c1-300 $OUT/bench_20.json
echo "== controller-step probes"
timeout 900 python tools/probe_step.py c2 rebal grbal mbmpc --calls=1000 > $OUT/probe_steps.jsonl 2> $OUT/probe.err; echo "probe rc=$?"
cut -c1-1200 $OUT/probe_steps.jsonl
timeout 900 python tools/probe_step.py c2 --calls=5000 > $OUT/probe_jitter.jsonl 2>> $OUT/probe.err; echo "jitter rc=$?"
cat $OUT/probe_jitter.jsonl
L2A_SYNC_SLEEP=0 timeout 900 python tools/probe_step.py c2 --calls=5000 > $OUT/probe_jitter_spin.jsonl 2>> $OUT/probe.err; echo "jitter spin rc=$?"
cut -c1-600 $OUT/probe_jitter_spin.jsonl
echo "== N = 2 code path: two gloo ranks sharing this GPU (self-launch; numbers meaningless)"
L2A_BENCH_SHARE_GPU=1 L2A_SPLIT=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_share2.json 2> $OUT/bench_share2.err; echo "share2 rc=$?"
cat $OUT/bench_share2.json; tail -5 $OUT/bench_share2.err
