#!/bin/bash
# per-dispatch timeline of one device-mode CEM plan step (config 5), developer aid: bash tools/cem_trace.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cat > /tmp/cem_run.py <<EOF
import sys, numpy as np, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import cases
case = cases.CASES["c5_hc_cem_n4000_h30_e5"]
ctrl = cases.product_controller(case, rng="device")
obs = cases.load_golden("c5_hc_cem_n4000_h30_e5_s0")["obs0"]
for _ in range(4):
    ctrl.get_actions(obs)
torch.cuda.synchronize()
EOF
cd /tmp && rm -rf /tmp/prof_cem
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cem -o p -- python /tmp/cem_run.py > /dev/null 2>&1
f=$(find /tmp/prof_cem -name "*kernel_trace.csv" | head -1)
python3 - <<EOF
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-26:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start %9.1f us  dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"][:44], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
EOF
