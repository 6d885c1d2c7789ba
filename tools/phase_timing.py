#!/usr/bin/env python3
"""Developer tool: per-phase s_memtime stamps of rq_forward_kernel (wave 0 of workgroup 0).
Build the instrumented library HERE (no GPU needed):   python tools/phase_timing.py --build
Run on the GPU box:                                    python tools/phase_timing.py B,D,K,L
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rq-vae-recommender_amd", "csrc")
SO = os.path.join(ROOT, "tools", "librqhip_timing.bin")

if len(sys.argv) > 1 and sys.argv[1] == "--build":
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-DRQ_TIMING", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", SO] + srcs
    subprocess.run(cmd, check=True)
    print("built", SO)
    sys.exit(0)

import torch  # noqa: E402

B, D, K, L = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "65536,32,256,3").split(","))
lib = C.CDLL(SO)
lib.rqhip_rq_forward_workspace_bytes.restype = C.c_size_t
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, D, generator=g) * 0.5).cuda()
cb = (torch.randn(L, K, D, generator=g) * 0.3).cuda()
ids = torch.empty((L, B), dtype=torch.int64, device="cuda")
es = torch.empty((B, D), device="cuda")
loss = torch.empty((B,), device="cuda")
norm = torch.empty((B, L), device="cuda")
wsb = lib.rqhip_rq_forward_workspace_bytes(L, K)
ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
vp = C.c_void_p
for _ in range(3):
    rc = lib.rqhip_rq_forward(vp(x.data_ptr()), C.c_int64(B), D, vp(cb.data_ptr()), L, K, 0, C.c_float(0.25),
                              vp(ids.data_ptr()), None, None, vp(es.data_ptr()), vp(loss.data_ptr()),
                              vp(norm.data_ptr()), vp(ws.data_ptr()), C.c_size_t(wsb), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.rqhip_debug_read(buf) == 0
t0 = buf[0]
names = {0: "kernel start", 1: "x loaded (after staging)", 100: "levels done", 101: "final stores issued"}
for l in range(L):
    names.update({2 + 8 * l: f"L{l} start", 3 + 8 * l: f"L{l} scan done", 4 + 8 * l: f"L{l} argmin merged",
                  5 + 8 * l: f"L{l} gather+loss done", 6 + 8 * l: f"L{l} output math done"})
prev = t0
for i in sorted(names):
    if buf[i]:
        print(f"{names[i]:28s} +{buf[i] - t0:8d} ticks  (delta {buf[i] - prev:7d})")
        prev = buf[i]
