export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/ab_double
for r in 1 2; do for d in 0 1; do
  L2A_DOUBLE=$d timeout 300 python tools/ab_kernel.py 2>/dev/null | sed "s/^{/{\"double\": $d, /" | tee -a gpurun_out/ab_double/ab_kernel.jsonl
  L2A_DOUBLE=$d timeout 300 python tools/ab_nt.py 2>/dev/null | sed "s/^{/{\"double\": $d, /" >> gpurun_out/ab_double/ab_nt.jsonl
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_shapes.py tests/test_native_step.py -m gpu -q -x --timeout 300 2>&1 | tail -4
