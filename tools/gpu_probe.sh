#!/bin/bash
# phase-skipping probes of rq_backward_flat_kernel (results are wrong with any bit set): 1 no table add, 2 no owner role,
# 4 no staging / barriers / owners, 8 no partial-table flush
O=gpurun_out/ab; mkdir -p $O
for pr in 0 1 2 4 12 8; do
  echo "== RQ_BWD_PROBE=$pr"
  RQ_BWD_PROBE=$pr timeout 100 python tools/bench_kernels.py bwd --one 1048576,32,256,3 --reps 30 --lib tools/_ab/librqhip_probe.so 2>&1 | grep "bwd ste"
  RQ_BWD_PROBE=$pr timeout 100 python tools/bench_kernels.py bwd --one 100000,32,256,3 --reps 30 --lib tools/_ab/librqhip_probe.so 2>&1 | grep "bwd ste"
done | tee $O/probe_bwd.log
python - <<'PY'
import torch, time
x = torch.randn(1048576, 32, device="cuda"); y = torch.empty_like(x); z = torch.randn_like(x)
for name, fn in (("copy 134MB", lambda: y.copy_(x)), ("add 2x134->134", lambda: torch.add(x, z, out=y))):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b) / 20 * 1e3, "us")
PY
