#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/thr
for round in 1 2 3; do
for t in 8 2 1; do
  L2A_RNG_THREADS=$t timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.readline()); c=b['config']
print(json.dumps({'rng_threads':$t,'value':b['value'],'ms':b['ms_per_step'],'kernel_ms':b['roofline']['kernel_ms'],'foreign':c['get_actions_parity_foreign_draw_plan_steps_per_s'],'device':c['get_actions_device_rng_plan_steps_per_s']}))" | tee -a gpurun_out/thr/threads.jsonl
done
done
