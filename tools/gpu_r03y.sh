#!/bin/bash
# round 3: cache policy of the exchange granules (sc1 | sc1 nt | sc0 sc1): kernel A/B, parity of the variants, L2 fetch
TAG=${TAG:-r03y}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
LIBD=$GRAFT_REPO_ROOT/learning_to_adapt_amd
for round in 1 2; do
  for v in "" _xnt _xntst _xsys; do
    L2A_LIB_PATH=$LIBD/libl2a_hip$v.so timeout 300 python tools/ab_kernel.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
    L2A_LIB_PATH=$LIBD/libl2a_hip$v.so timeout 300 python tools/ab_lstm.py 2>> $OUT/ab.err | tee -a $OUT/ab.jsonl
  done
done
for v in _xnt _xsys; do
  L2A_LIB_PATH=$LIBD/libl2a_hip$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_rnn.py -m gpu -q --timeout 300 -x -k "split or batching or golden" > $OUT/pytest$v.log 2>&1; echo "pytest $v rc=$?"; tail -2 $OUT/pytest$v.log
done
for v in "" _xnt; do
  for set in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && L2A_LIB_PATH=$LIBD/libl2a_hip$v.so timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc${v}_$set -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc$v.err); echo "pmc $v $set rc=$?"
  python - <<PY
import csv, glob
f = glob.glob("$OUT/pmc${v}_$set/**/*counter_collection.csv", recursive=True)
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "rollout_mfma" in r["Kernel_Name"] and r["Counter_Name"] == "$set"]
print("$v $set per launch (KB): mean %.0f over %d" % (sum(vals) / len(vals), len(vals)))
PY
  done
done
