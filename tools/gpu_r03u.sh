#!/bin/bash
# round 3: timelines of the unsplit five-set shapes (c5: 250 tiles NT=1; c4: NT=2) - is the GEMM phase slowed by L2 misses?
TAG=${TAG:-r03u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for c in c5_hc_cem_n4000_h30_e5 c4_hc_rs_n16000_h30_e5 c3_ant_rs_n2000_h20_pb5; do
  timeout 200 python tools/timeline.py $c > $OUT/timeline_$c.txt 2>&1; echo "timeline $c rc=$?"
  grep -v amdgpu $OUT/timeline_$c.txt | cut -c1-330 | sed -n 1,42p
done
