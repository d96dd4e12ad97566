#!/bin/bash
# per-dispatch timeline of the adaptation step's launches (developer aid): bash tools/adapt_trace.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rm -rf /tmp/prof_adapt
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_adapt -o p -- python $R/tools/probe_adapt_prof.py > /dev/null 2>&1
f=$(find /tmp/prof_adapt -name "*kernel_trace.csv" | head -1)
python3 - <<EOF
import csv
rows = [r for r in csv.DictReader(open("$f")) if "adapt" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 8 if len(rows) % 8 == 0 else 12
base = rows[-n:]
t0 = int(base[0]["Start_Timestamp"])
for r in base:
    print("%-24s start %7.1f us  dur %6.1f us  grid %sx%s wg %s" % (r["Kernel_Name"][:22], (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"]))
print("step %.1f us" % ((int(base[-1]["End_Timestamp"]) - t0) / 1e3))
EOF
