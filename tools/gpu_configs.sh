#!/bin/bash
# configs table + full GPU suite with the current library
TAG=${TAG:-cfg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/%s/configs.jsonl" % __import__("os").environ.get("TAG", "cfg")):
    r = json.loads(l)
    print(r["config"][:90], {k: r[k] for k in ("kernel_ms", "frac_fp32_peak", "ms_per_call", "step_ms", "env_steps_per_s") if k in r})
PY
