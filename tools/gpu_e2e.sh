#!/bin/bash
# end-to-end controller steps of the reference's own run-script defaults + stage probes
TAG=${TAG:-e2e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "payload or golden" 2>&1 | tail -2
timeout 900 python tools/bench_configs.py 2> $OUT/configs.err | grep "controller step\|end to end" | tee $OUT/configs_e2e.jsonl | cut -c1-260
echo "== stage probes"
timeout 300 python tools/probe_e2e.py c6_hc_rnn_rs_n500_h10_m5 2>/dev/null | tee $OUT/probe_c6.jsonl | cut -c1-400
timeout 300 python tools/probe_e2e.py c3b_ant_rs_n500_h10_pb5_3x512 2>/dev/null | tee $OUT/probe_c3b.jsonl | cut -c1-400
timeout 300 python tools/probe_e2e.py 2>/dev/null | tee $OUT/probe_c2.jsonl | cut -c1-400
