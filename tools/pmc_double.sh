#!/bin/bash
# PMC passes (counters only, own runs) for the double rounds: BASELINE config 3 (front launch on double tiles + the rest on micro
# tiles) and config 4's shape on one GPU (all on double tiles), double rounds on and off (tools/prof_defaults.py c3 | c4).
#   gpurun --timeout 900 -- 'bash tools/pmc_double.sh r06'
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_pmc_double
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for shape in c3 c4; do for dbl in 1 0; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $set | cut -d" " -f1)
    (cd /tmp && L2A_DOUBLE=$dbl timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${shape}${dbl}_$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_defaults.py $shape 20 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_${shape}${dbl}_$name.err); echo "pmc $shape double=$dbl $name rc=$?"
  done
  for k in "l2a_rollout_mfma_k<2" "l2a_rollout_mfma_k<1" "l2a_mlp_micro_k"; do
    grep -q "$k" $OUT/pmc_${shape}${dbl}_SQ_WAVES/pmc_counter_collection.csv 2>/dev/null && (echo "# $shape (tools/prof_defaults.py $shape 20), L2A_DOUBLE=$dbl: $k"; python tools/pmc_summary.py $OUT "$k" pmc_${shape}${dbl}_) >> $OUT/pmc_double.txt 2>&1
  done
done; done
cat $OUT/pmc_double.txt
find $OUT -name "*.db" -delete 2>/dev/null
