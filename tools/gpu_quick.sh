#!/bin/bash
# Short GPU-box session for work in progress: the named pytest selections, the controller-step probe, one bench line.
#   gpurun --timeout 900 -- 'bash tools/gpu_quick.sh r05a "tests/test_native_step.py" "c2 rebal grbal mbmpc"'
TAG=${1:-quick}
SEL=${2:-tests/test_native_step.py}
WL=${3:-c2 rebal grbal mbmpc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
lscpu | grep -E "Model name|^CPU\(s\)" > $OUT/host_cpu.txt
echo "== pytest $SEL"
timeout 900 python -m pytest $SEL -m gpu -q -x --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
echo "== probe_step $WL"
timeout 600 python tools/probe_step.py $WL > $OUT/probe_steps.jsonl 2> $OUT/probe_steps.err; echo "probe rc=$?"
cat $OUT/probe_steps.jsonl; tail -5 $OUT/probe_steps.err
echo "== same with the Python path (L2A_NATIVE_STEP=0)"
L2A_NATIVE_STEP=0 timeout 600 python tools/probe_step.py $WL --calls=300 > $OUT/probe_steps_python.jsonl 2>> $OUT/probe_steps.err; echo "probe rc=$?"
cut -c1-400 $OUT/probe_steps_python.jsonl
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
