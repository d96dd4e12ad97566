#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== 2 ranks sharing the GPU over gloo (code-path exercise of --gpus 2 incl. self-launch; numbers meaningless)"
L2A_BENCH_SHARE_GPU=1 L2A_SPLIT=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_share2.json 2> $OUT/bench_share2.err; echo "rc=$?"
cat $OUT/bench_share2.json; tail -15 $OUT/bench_share2.err
echo "== probe cem"
timeout 300 python tools/probe_e2e.py c5_hc_cem_n4000_h30_e5 > $OUT/probe_cem.jsonl 2> $OUT/probe_cem.err; cat $OUT/probe_cem.jsonl; tail -5 $OUT/probe_cem.err
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
