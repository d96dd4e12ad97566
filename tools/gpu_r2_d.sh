#!/bin/bash
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
echo "== probe cem"
timeout 300 python tools/probe_e2e.py c5_hc_cem_n4000_h30_e5 > $OUT/probe_cem.jsonl 2> $OUT/probe_cem.err; cat $OUT/probe_cem.jsonl; tail -5 $OUT/probe_cem.err
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
