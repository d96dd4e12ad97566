#!/bin/bash
# GPU-box session for the member fan (round 6): phase timeline of one rank's config-5 shard, rocprofv3 kernel statistics of the
# shard launch with the fan on and off, the shard probe, one bench line.
#   gpurun --timeout 900 -- 'bash tools/gpu_fan.sh r06b'
TAG=${1:-r06fan}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== timeline (fan, n=500)"
timeout 200 python tools/timeline.py c5_hc_cem_n4000_h30_e5 n=500 > $OUT/timeline_c5shard.txt 2>&1; echo "timeline rc=$?"
grep -v amdgpu.ids $OUT/timeline_c5shard.txt | cut -c1-400
echo "== rocprof kernel statistics of the shard launch"
for fan in 1 0; do
  (cd /tmp && L2A_FAN=$fan timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c5 -o trace -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py c5shard 400 > /dev/null 2>> $GRAFT_REPO_ROOT/$OUT/prof_c5.err)
  f=$(find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (echo "# config 5 shard (n=500, h=30, E=5 mean; tools/prof_defaults.py c5shard 400), L2A_FAN=$fan"; head -1 "$f"; grep -E '^"(void )?l2a_(rollout|mlp)' "$f") >> $OUT/c5shard_kernel_stats.csv
  rm -rf $OUT/prof_c5
done
cat $OUT/c5shard_kernel_stats.csv | cut -c1-200
echo "== shard probe"
timeout 300 python tools/probe_c5_shard.py > $OUT/probe_c5_shard.jsonl 2> $OUT/probe.err; echo "probe rc=$?"
cat $OUT/probe_c5_shard.jsonl
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
