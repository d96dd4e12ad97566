#!/bin/bash
# Per-dispatch timeline of the LAST controller steps of a default workload (developer aid): bash tools/step_trace.sh rebal|grbal|c2|mbmpc
# rocprofv3 --kernel-trace over tools/probe_step.py <workload> --calls=60; prints the launches of the final three steps.
export TMPDIR=/tmp
W=${1:-rebal}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rm -rf /tmp/prof_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o p -- python $R/tools/probe_step.py $W --calls=60 > /dev/null 2>&1
f=$(find /tmp/prof_step -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("$f"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = the launches between two rollout kernels (the longest kernel of the workload)
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
names = {}
for r in rows:
    names.setdefault(r["Kernel_Name"], []).append(dur(r))
main = max(names, key=lambda k: sum(names[k]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"] == main]
# the un-instrumented pass of probe_step comes first (calls), then the event pass: take steps from the middle of the first pass
mid = idx[len(idx) // 3]
lo = idx[idx.index(mid) - 1] + 1
hi = idx[idx.index(mid) + 2]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else "  (idle before: %5.1f us)" % ((s - prev_end) / 1e3)
    print("%-46s start %8.1f us  dur %7.1f us  grid %6sx%-3s wg %s%s" % (r["Kernel_Name"][:46], (s - t0) / 1e3, (e - s) / 1e3,
          r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], gap))
    prev_end = e
PY
