#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprof kernel trace.  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01'
# Everything lands under gpurun_out/<tag>/ (scratch; copy what should be judged into profiles/).
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/host_cpu.txt
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -5 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
if [ "${PMC:-1}" = "1" ]; then
echo "== rocprof PMC passes (counters only, own runs)"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d" " -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.err); echo "pmc $name rc=$?"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
fi
echo "== configs / parity report / timelines"
timeout 600 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"
timeout 300 python tools/parity_report.py > $OUT/parity_report.txt 2> $OUT/parity.err; echo "parity rc=$?"
timeout 120 python tools/timeline.py > $OUT/timeline.txt 2>&1; echo "timeline rc=$?"
timeout 120 python tools/timeline_lstm.py 256 4096 > $OUT/timeline_lstm.txt 2>&1; echo "timeline_lstm rc=$?"
timeout 120 python tools/timeline_lstm.py 256 2000 >> $OUT/timeline_lstm.txt 2>&1; echo "timeline_lstm (split) rc=$?"
echo "== micro-tile kernels: timelines, A/B against the 16-candidate kernels, rocprof kernel statistics of the default plan sizes"
for a in "lstm 500 5 10" "mlp c3b_ant_rs_n500_h10_pb5_3x512" "mlp c1_hc_rs_n500_h10_e1" "mlp c2_hc_rs_n2000_h30_e5"; do
  n=$(echo $a | tr ' ' '_'); timeout 200 python tools/timeline_micro.py $a > $OUT/timeline_micro_$n.txt 2>&1; echo "timeline_micro $a rc=$?"
done
for a in "gru 256" "lstm 256,256" "rnn 256"; do
  n=$(echo $a | tr ' ,' '_x'); timeout 200 python tools/timeline_micro.py rnn $a > $OUT/timeline_micro_rnn_$n.txt 2>&1; echo "timeline_micro rnn $a rc=$?"
done
timeout 400 python tools/ab_rnn_micro.py > $OUT/ab_rnn_micro.jsonl 2> $OUT/ab_rnn_micro.err; echo "ab_rnn_micro rc=$?"
timeout 500 python tools/ab_micro.py lstm mlp > $OUT/ab_micro.jsonl 2> $OUT/ab_micro.err; echo "ab_micro rc=$?"
timeout 300 python tools/ab_nt.py > $OUT/ab_nt.jsonl 2> /dev/null; echo "ab_nt rc=$?"
for shape in c1 c3b c6 gru lstm2; do for mic in 1 0; do
  (cd /tmp && L2A_MICRO=$mic timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_def -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py $shape > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/prof_def.err)
  f=$(find $OUT/prof_def -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (echo "# $shape, L2A_MICRO=$mic"; head -1 "$f"; grep -E '^"(void )?l2a_(rollout|lstm|mlp|rnn)' "$f") >> $OUT/defaults_kernel_stats.csv
  rm -rf $OUT/prof_def
done; done; echo "defaults rocprof done"; cat $OUT/defaults_kernel_stats.csv | cut -c1-150
echo "== PMC passes for the micro-tile kernels (GRU 256, LSTM 2 x 256, c3b, c6, c1 at their default plan sizes; counters only, own runs)"
for shape in gru lstm2 c3b c6 c1; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $set | cut -d" " -f1)
    (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmcr_${shape}_$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py $shape 20 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmcr_${shape}_$name.err); echo "pmc $shape $name rc=$?"
  done
  case $shape in gru|lstm2) kn=l2a_rnn_micro_k; of=pmc_rnn_micro.txt;; c6) kn=l2a_lstm_micro_k; of=pmc_micro.txt;; *) kn=l2a_mlp_micro_k; of=pmc_micro.txt;; esac
  (echo "# $shape (tools/prof_defaults.py $shape 20): $kn"; python tools/pmc_summary.py $OUT $kn pmcr_${shape}_) >> $OUT/$of 2>&1
done; cat $OUT/pmc_rnn_micro.txt $OUT/pmc_micro.txt
echo "== GrBAL adaptation step"
timeout 120 python tools/probe_adapt.py 2> /dev/null > $OUT/probe_adapt.json; echo "probe_adapt rc=$?"
bash tools/adapt_trace.sh > $OUT/adapt_trace.txt 2>&1; echo "adapt_trace rc=$?"; cd $GRAFT_REPO_ROOT
bash tools/cem_trace.sh > $OUT/cem_trace.txt 2>&1; echo "cem_trace rc=$?"; cd $GRAFT_REPO_ROOT
echo "== round 5: the controller step as one C call - stage tables of the four default workloads, the 5000-call distribution, dispatch traces"
timeout 900 python tools/probe_step.py c2 rebal grbal mbmpc --calls=1000 > $OUT/probe_steps.jsonl 2> $OUT/probe_steps.err; echo "probe_steps rc=$?"
timeout 600 python tools/probe_step.py c2 --calls=5000 > $OUT/probe_jitter.jsonl 2>> $OUT/probe_steps.err; echo "probe_jitter rc=$?"
L2A_NATIVE_STEP=0 timeout 600 python tools/probe_step.py c2 rebal grbal mbmpc --calls=500 > $OUT/probe_steps_python_path.jsonl 2>> $OUT/probe_steps.err; echo "probe_steps (python path) rc=$?"
for w in c2 rebal grbal mbmpc; do bash tools/step_trace.sh $w > $OUT/step_trace_$w.txt 2>&1; cd $GRAFT_REPO_ROOT; done; echo "step traces done"
timeout 120 python tools/timeline_adapt.py cold > $OUT/timeline_adapt.txt 2>&1; echo "timeline_adapt rc=$?"
timeout 300 python tools/two_planners.py > $OUT/two_planners.jsonl 2> $OUT/two_planners.err; echo "two_planners rc=$?"
timeout 300 python tools/soak_step.py 6000 > $OUT/soak_step.txt 2>&1; echo "soak_step rc=$?"
timeout 600 python tools/soak.py > $OUT/soak.txt 2>&1; echo "soak rc=$?"
echo "== host RNG helper / end-to-end stage probes"
timeout 300 python tools/bench_rng.py > $OUT/rng.jsonl 2>&1; echo "rng rc=$?"
timeout 300 python tools/probe_e2e.py > $OUT/probe_c2.jsonl 2> $OUT/probe.err; echo "probe c2 rc=$?"
timeout 300 python tools/probe_e2e.py c5_hc_cem_n4000_h30_e5 > $OUT/probe_c5.jsonl 2>> $OUT/probe.err; echo "probe c5 rc=$?"
echo "== N = 2 code path: two gloo ranks sharing this GPU (self-launch; numbers meaningless)"
L2A_BENCH_SHARE_GPU=1 L2A_SPLIT=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_share2.json 2> $OUT/bench_share2.err; echo "share2 rc=$?"
echo "== round 6: member fan (config-5 shards), the stationary-cluster probe, two planners on the blocking path"
timeout 200 python tools/timeline.py c5_hc_cem_n4000_h30_e5 n=500 > $OUT/timeline_c5shard_fan.txt 2>&1; echo "timeline fan rc=$?"
timeout 200 python tools/timeline.py c3_ant_rs_n2000_h20_pb5 > $OUT/timeline_c3.txt 2>&1; echo "timeline c3 (rest launch: shared tiles) rc=$?"
L2A_DBG_FRONT=1 timeout 200 python tools/timeline.py c3_ant_rs_n2000_h20_pb5 > $OUT/timeline_c3_double.txt 2>&1; echo "timeline c3 (front launch: double tiles) rc=$?"
L2A_DOUBLE=0 timeout 200 python tools/timeline.py c3_ant_rs_n2000_h20_pb5 > $OUT/timeline_c3_single.txt 2>&1; echo "timeline c3 (double rounds off) rc=$?"
for fan in 1 0; do
  (cd /tmp && L2A_FAN=$fan timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c5 -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py c5shard 400 > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/prof_c5.err)
  f=$(find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (echo "# config 5 shard (n=500, h=30, E=5 mean; tools/prof_defaults.py c5shard 400), L2A_FAN=$fan"; head -1 "$f"; grep -E '^"(void )?l2a_(rollout|mlp)' "$f") >> $OUT/c5shard_kernel_stats.csv
  rm -rf $OUT/prof_c5
done; cat $OUT/c5shard_kernel_stats.csv | cut -c1-160
for shape in c3 c4; do for dbl in 1 0; do
  (cd /tmp && L2A_DOUBLE=$dbl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_dbl -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py $shape 100 > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/prof_dbl.err)
  f=$(find $OUT/prof_dbl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (echo "# $shape (tools/prof_defaults.py $shape 100), L2A_DOUBLE=$dbl"; head -1 "$f"; grep -E '^"(void )?l2a_(rollout|mlp)' "$f") >> $OUT/double_kernel_stats.csv
  rm -rf $OUT/prof_dbl
done; done; cat $OUT/double_kernel_stats.csv | cut -c1-170
for dbl in 0 1; do for r in 1 2; do L2A_DOUBLE=$dbl timeout 300 python tools/ab_kernel.py 2>/dev/null | sed "s/^{/{\"double\": $dbl, /" >> $OUT/ab_double.jsonl; done; done; echo "ab_double rc=$?"
timeout 300 python tools/probe_c5_shard.py > $OUT/probe_c5_shard.jsonl 2> $OUT/probe_c5_shard.err; echo "probe_c5_shard rc=$?"
[ -x tools/probes/stationary ] && (timeout 120 tools/probes/stationary 10 > $OUT/probe_stationary.jsonl 2>&1; echo "stationary rc=$?")
timeout 300 python tools/two_planners.py sync > $OUT/two_planners_sync.jsonl 2> /dev/null; echo "two_planners sync rc=$?"
# keep the merged-back payload small
find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null
du -sh $OUT
