#!/bin/bash
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== probe cem"
timeout 300 python tools/probe_e2e.py c5_hc_cem_n4000_h30_e5 > $OUT/probe_cem.jsonl 2> $OUT/probe_cem.err; cat $OUT/probe_cem.jsonl; tail -5 $OUT/probe_cem.err
echo "== rng"
timeout 300 python tools/bench_rng.py > $OUT/rng.jsonl 2>&1; cat $OUT/rng.jsonl
echo "== cpu tests of the rng on this host"
timeout 600 python -m pytest tests/test_rng_draw_ahead.py -q 2>&1 | tail -3
echo "== timeline"
timeout 120 python tools/timeline.py > $OUT/timeline.txt 2>&1; tail -12 $OUT/timeline.txt
