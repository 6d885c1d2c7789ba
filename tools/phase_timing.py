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

B, D, K, L, *rest = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "65536,32,256,3").split(","))
MODE = rest[0] if rest else 0      # 0 eval / 1 STE / 2 rotation trick (B,D,K,L,mode)
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
    rc = lib.rqhip_rq_forward(vp(x.data_ptr()), C.c_int64(B), D, vp(cb.data_ptr()), L, K, MODE, C.c_float(0.25),
                              vp(ids.data_ptr()), None, None, vp(es.data_ptr()), vp(loss.data_ptr()),
                              vp(norm.data_ptr()), None, vp(ws.data_ptr()), C.c_size_t(wsb), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.rqhip_debug_read(buf) == 0
t0 = buf[0]
names = {200: "rows requested", 201: "staging issued (own part)", 202: "staging barrier passed", 0: "kernel start", 1: "x loaded (after staging)", 100: "levels done", 101: "final stores issued"}
for l in range(L):
    names.update({2 + 8 * l: f"L{l} start", 3 + 8 * l: f"L{l} scan done", 4 + 8 * l: f"L{l} argmin merged",
                  5 + 8 * l: f"L{l} gather+loss done", 6 + 8 * l: f"L{l} output math done"})
prev = t0
for i in sorted(names, key=lambda i: buf[i]):
    if buf[i]:
        print(f"{names[i]:28s} +{buf[i] - t0:8d} ticks  (delta {buf[i] - prev:7d})")
        prev = buf[i]

# ---- whole-kernel picture: per-wave trace (100 MHz clock -> 10 ns ticks) --------------------------------
import numpy as np  # noqa: E402

lib.rqhip_debug_trace(None, 1)
torch.cuda.synchronize()
rc = lib.rqhip_rq_forward(vp(x.data_ptr()), C.c_int64(B), D, vp(cb.data_ptr()), L, K, MODE, C.c_float(0.25),
                          vp(ids.data_ptr()), None, None, vp(es.data_ptr()), vp(loss.data_ptr()),
                          vp(norm.data_ptr()), None, vp(ws.data_ptr()), C.c_size_t(wsb), None)
torch.cuda.synchronize()
tr = np.zeros((4096, 16, 8), np.uint64)
assert lib.rqhip_debug_trace(tr.ctypes.data_as(C.c_void_p), 0) == 0
used = tr[:, :, 0] > 0
t0 = tr[:, :, 0][used].min()
rel = (tr.astype(np.int64) - np.int64(t0)) / 100.0  # microseconds
rel[tr == 0] = np.nan
print(f"\nper-wave trace, {int(used.sum())} waves in {int(used.any(axis=1).sum())} workgroups (us since first wave entered):")
for slot, name in ((0, "kernel entry"), (1, "codebooks staged"), (2, "tile 1 done"), (3, "tile 2 done"), (4, "tile 3 done")):
    v = rel[:, :, slot][used & (tr[:, :, slot] > 0)]
    if v.size:
        q = np.percentile(v, [0, 10, 50, 90, 100])
        print(f"  {name:18s} n={v.size:6d}  min {q[0]:7.2f}  p10 {q[1]:7.2f}  median {q[2]:7.2f}  p90 {q[3]:7.2f}  max {q[4]:7.2f}")
d1 = rel[:, :, 2] - rel[:, :, 1]
print("  tile-1 duration by wave index within the workgroup (median us):",
      " ".join(f"{np.nanmedian(d1[:, w]):.1f}" for w in range(16) if used[:, w].any()))
d2 = rel[:, :, 3] - rel[:, :, 2]
if np.isfinite(d2).any():
    print(f"  tile-2 duration: n={int(np.isfinite(d2).sum())} median {np.nanmedian(d2):.1f} us, max {np.nanmax(d2):.1f} us")
