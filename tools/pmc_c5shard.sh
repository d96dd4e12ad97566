#!/bin/bash
# PMC passes (counters only, own runs) for the member-fan launch of one rank's config-5 shard (tools/prof_defaults.py c5shard),
# fan on and off.   gpurun --timeout 600 -- 'bash tools/pmc_c5shard.sh r06'
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_pmc_c5
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for fan in 1 0; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $set | cut -d" " -f1)
    (cd /tmp && L2A_FAN=$fan timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc${fan}_$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py c5shard 20 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc${fan}_$name.err); echo "pmc fan=$fan $name rc=$?"
  done
  (echo "# config 5 shard (n=500, h=30, E=5 mean; tools/prof_defaults.py c5shard 20), L2A_FAN=$fan: l2a_rollout_mfma_k"; python tools/pmc_summary.py $OUT l2a_rollout pmc${fan}_) >> $OUT/pmc_c5shard.txt 2>&1
done
cat $OUT/pmc_c5shard.txt
find $OUT -name "*.db" -delete 2>/dev/null
