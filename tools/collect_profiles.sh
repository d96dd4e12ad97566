#!/bin/bash
# Copy the judged artefacts of one GPU round (gpurun_out/<tag>/, scratch) into profiles/ (tracked).
#   bash tools/collect_profiles.sh r02
TAG=${1:-r02}
SRC=gpurun_out/$TAG
DST=profiles
cp $SRC/bench.json $DST/${TAG}_bench.json
cp $SRC/bench_share2.json $DST/${TAG}_bench_share2_gloo.json
cp $SRC/configs.jsonl $DST/${TAG}_configs.jsonl
cp $SRC/host_cpu.txt $DST/${TAG}_host_cpu.txt
cp $SRC/pmc_summary.txt $DST/${TAG}_pmc_summary.txt
cp $SRC/probe_c2.jsonl $DST/${TAG}_probe_c2.jsonl
cp $SRC/probe_c5.jsonl $DST/${TAG}_probe_c5.jsonl
cp $SRC/rng.jsonl $DST/${TAG}_rng.jsonl
cp $SRC/smoke.log $DST/${TAG}_smoke.log
grep -v amdgpu.ids $SRC/timeline.txt > $DST/${TAG}_timeline.txt
grep -v amdgpu.ids $SRC/timeline_lstm.txt > $DST/${TAG}_timeline_lstm.txt
cp $SRC/parity_report.txt $DST/${TAG}_parity_report.txt
cp $SRC/probe_adapt.json $DST/${TAG}_probe_adapt.json
grep -v amdgpu.ids $SRC/adapt_trace.txt > $DST/${TAG}_adapt_trace.txt
grep -v amdgpu.ids $SRC/cem_trace.txt | cut -c1-140 > $DST/${TAG}_cem_trace.txt
[ -f $SRC/probe_jitter.json ] && cp $SRC/probe_jitter.json $DST/${TAG}_probe_jitter.json
for f in probe_steps.jsonl probe_jitter.jsonl probe_steps_python_path.jsonl two_planners.jsonl; do [ -f $SRC/$f ] && cp $SRC/$f $DST/${TAG}_$f; done
for f in $SRC/step_trace_*.txt; do [ -f "$f" ] && grep -v amdgpu.ids $f | cut -c1-150 > $DST/${TAG}_$(basename $f); done
[ -f $SRC/soak_step.txt ] && grep -v amdgpu.ids $SRC/soak_step.txt > $DST/${TAG}_soak_step.txt
[ -f $SRC/soak.txt ] && grep -v amdgpu.ids $SRC/soak.txt | tail -30 > $DST/${TAG}_soak.txt
[ -f $SRC/timeline_adapt.txt ] && grep -v amdgpu.ids $SRC/timeline_adapt.txt > $DST/${TAG}_timeline_adapt.txt
tail -12 $SRC/pytest_gpu.log | grep -v amdgpu.ids > $DST/${TAG}_pytest_gpu_tail.txt
# rocprofv3 --stats: the rollout kernels only (torch's elementwise kernels have kilobyte-long names)
f=$(find $SRC/prof -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; grep -E '^"(void )?l2a_' "$f") > $DST/${TAG}_kernel_stats.csv
for f in $SRC/timeline_micro_*.txt; do [ -f "$f" ] && grep -v amdgpu.ids $f > $DST/${TAG}_$(basename $f); done
[ -f $SRC/ab_micro.jsonl ] && cp $SRC/ab_micro.jsonl $DST/${TAG}_ab_micro.jsonl
[ -f $SRC/ab_nt.jsonl ] && cp $SRC/ab_nt.jsonl $DST/${TAG}_ab_nt.jsonl
[ -f $SRC/pmc_rnn_micro.txt ] && cp $SRC/pmc_rnn_micro.txt $DST/${TAG}_pmc_rnn_micro.txt
[ -f $SRC/pmc_micro.txt ] && cp $SRC/pmc_micro.txt $DST/${TAG}_pmc_micro.txt
[ -f $SRC/ab_rnn_micro.jsonl ] && grep -v amdgpu.ids $SRC/ab_rnn_micro.jsonl > $DST/${TAG}_ab_rnn_micro.jsonl
[ -f $SRC/defaults_kernel_stats.csv ] && cp $SRC/defaults_kernel_stats.csv $DST/${TAG}_defaults_kernel_stats.csv
for f in c5shard_kernel_stats.csv double_kernel_stats.csv probe_c5_shard.jsonl two_planners_sync.jsonl; do [ -f $SRC/$f ] && cp $SRC/$f $DST/${TAG}_$f; done
for f in timeline_c5shard_fan.txt timeline_c3.txt timeline_c3_double.txt timeline_c3_single.txt; do [ -f $SRC/$f ] && grep -v amdgpu.ids $SRC/$f | cut -c1-420 > $DST/${TAG}_$f; done
python tools/pmc_traffic.py $SRC > /dev/null 2>&1 || true
ls -la $DST | grep ${TAG}_ | awk '{print $5, $9}'
