#!/bin/bash
# A/B of member-fan kernel variants on one GPU box (libraries built by tools/build_variant.py <name> --units=l2a_mfma_fan_1_8 -D...):
#   gpurun --timeout 600 -- 'bash tools/ab_fan.sh base fanpf fanpub fanboth'
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/ab_fan
for v in "$@"; do
  lib=""; [ "$v" != "base" ] && lib=$GRAFT_REPO_ROOT/learning_to_adapt_amd/libl2a_hip_$v.so
  for rep in 1 2; do
    L2A_LIB_PATH=$lib timeout 120 python - <<PY 2>/dev/null | tee -a gpurun_out/ab_fan/ab_fan.jsonl
import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import torch, cases
from bench_configs import time_launches
full = cases.CASES["c5_hc_cem_n4000_h30_e5"]
out = {"variant": "$v", "rep": $rep}
for n in (500, 1000):
    case = dict(full, n=n)
    env, model = cases.product_model(case)
    native = model.planner_model(); dev = native.device
    obs0 = torch.randn((1, 20), device=dev); a = torch.rand((30, n, 6), device=dev) * 2 - 1
    best = torch.zeros((1,), dtype=torch.int64, device=dev); rets = torch.zeros((1, n), dtype=torch.float32, device=dev)
    out["n%d_ms" % n] = round(time_launches(lambda: native.plan_rs(obs0, a, 1, n, 30, 1.0, env.reward_spec, returns_out=rets, best_key=best), 60), 4)
    native.ctx.launch_status()
    out["n%d_checksum" % n] = float(rets.double().sum().cpu())
print(json.dumps(out))
PY
  done
done
