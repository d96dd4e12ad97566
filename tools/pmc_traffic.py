#!/usr/bin/env python
"""profiles/pmc_traffic.json (read by bench.py for `roofline.traffic`) from a round's PMC summary.

    python tools/pmc_traffic.py gpurun_out/r02 [tag]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.normpath(src))
text = open(os.path.join(src, "pmc_summary.txt")).read()
fetch = float(re.search(r"FETCH_SIZE per launch = ([0-9.]+) KB", text).group(1))
write = float(re.search(r"WRITE_SIZE per launch = ([0-9.]+) KB", text).group(1))
hit = float(re.search(r"L2 hit rate = ([0-9.]+)", text).group(1))
total = int(round((2.0 * fetch + write) * 1024))
out = {
    "hbm_bytes_per_launch": total,
    "source": "profiles/%s_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py "
              "--steps 5 --no-e2e), mean per launch of l2a_rollout_mfma_k; FETCH_SIZE doubled per MI355X_MICROARCH.md "
              "(gfx950 counts 128-B requests at 64 B); memory-side (Infinity-Cache-inclusive) bytes: ~%d MB fetched by the 8 XCD "
              "L2s (%.1f %% L2 hit rate) = the partner's exchange granules, read with sc1 (>= the %d MB written) + each "
              "L2's first touch of its weight sets (8 x 2.9 MB) + re-fetches; + %d MB of write-through exchange granules; "
              "algorithmic bytes are 7.17 MB" % (tag, round(2 * fetch / 1000), 100 * hit, round(write / 1000), round(write / 1000)),
    "fetch_kb_raw": fetch,
    "write_kb": write,
}
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
