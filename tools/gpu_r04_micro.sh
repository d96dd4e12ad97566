#!/bin/bash
# round 4: micro-tile kernels - bit identity + timings (A/B by l2a_set_micro) + timelines; FULL=1: then the whole GPU suite
TAG=${TAG:-r04m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -k "micro" > $OUT/pytest_micro.log 2>&1; echo "pytest micro rc=$?"; tail -${TAILN:-25} $OUT/pytest_micro.log
timeout 400 python tools/ab_micro.py ${AB_WHAT:-lstm mlp} > $OUT/ab_micro.jsonl 2> $OUT/ab_micro.err; echo "ab rc=$?"; cat $OUT/ab_micro.jsonl; tail -3 $OUT/ab_micro.err
if [ -f learning_to_adapt_amd/libl2a_hip_timeline.so ]; then
timeout 200 python tools/timeline_micro.py lstm 500 5 10 > $OUT/tl_micro_lstm_c6.txt 2>&1; cat $OUT/tl_micro_lstm_c6.txt | grep -v amdgpu.ids
timeout 200 python tools/timeline_micro.py mlp c3b_ant_rs_n500_h10_pb5_3x512 > $OUT/tl_micro_mlp_c3b.txt 2>&1; cat $OUT/tl_micro_mlp_c3b.txt | grep -v amdgpu.ids
timeout 200 python tools/timeline_micro.py mlp c1_hc_rs_n500_h10_e1 > $OUT/tl_micro_mlp_c1.txt 2>&1; cat $OUT/tl_micro_mlp_c1.txt | grep -v amdgpu.ids
timeout 200 python tools/timeline_micro.py mlp c2_hc_rs_n2000_h30_e5 > $OUT/tl_micro_mlp_c2.txt 2>&1; cat $OUT/tl_micro_mlp_c2.txt | grep -v amdgpu.ids
fi
if [ "${FULL:-0}" = "1" ]; then
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
fi
