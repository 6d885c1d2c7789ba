#!/bin/bash
# Developer A/B: the Gumbel kernels with the library's logf / expf (RQ_GUMBEL_LIBM=1) instead of v_log_f32 / v_exp_f32.
# Build HERE:  bash tools/gumbel_libm_ab.sh build      -> tools/_ab/librqhip_libm.so
# GPU box:     bash tools/gumbel_libm_ab.sh run        (times both builds, prints each one's error against the reference goldens)
set -e
cd "$(dirname "$0")/.."
C=rq-vae-recommender_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/_ab; make -s -C $C
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -Iinclude -I$C -DRQ_GUMBEL_LIBM=1"
  for f in gumbel gumbel_mfma; do /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o tools/_ab/${f}_libm.o; done
  OBJS=$(ls $C/*.o | grep -v "/gumbel.o" | grep -v "/gumbel_mfma.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/librqhip_libm.so tools/_ab/gumbel_libm.o tools/_ab/gumbel_mfma_libm.o $OBJS
  rm -f tools/_ab/*_libm.o; echo built tools/_ab/librqhip_libm.so; exit 0
fi
for lib in - tools/_ab/librqhip_libm.so; do
python - $lib <<'PY'
import os, sys, time
import numpy as np, torch
ROOT = os.getcwd(); sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib
if sys.argv[1] != "-": _lib.load(os.path.abspath(sys.argv[1]))
from rqhip import ops
print("== library:", "product (v_log_f32 / v_exp_f32)" if sys.argv[1] == "-" else "libm logf / expf")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
for B, D, K in ((100000, 32, 256), (100000, 64, 256), (8192, 32, 256)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, D, device="cuda", generator=g) * 0.5; cb = torch.randn(K, D, device="cuda", generator=g) * 0.3
    U = torch.rand(B, K, device="cuda", generator=g); ge = torch.randn(B, D, device="cuda", generator=g); gl = torch.full((B,), 1.0 / B, device="cuda")
    f = timed(lambda: ops.gumbel_forward(x, cb, U, 0.2, 0.25)); b = timed(lambda: ops.gumbel_backward(x, cb, U, 0.2, 0.25, g_emb=ge, g_loss=gl))
    print(f"  B={B} D={D} K={K}: fwd {f:8.1f} us  bwd {b:8.1f} us")
# error against the reference's outputs (tests/golden/gumbel_*.npz; the small-batch kernels) and against an fp64 evaluation at 20 000 rows (matrix path)
for name in ("gumbel_a.npz", "gumbel_b.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    x, cb, U = (torch.from_numpy(g[k]).cuda() for k in ("x", "codebook", "U"))
    ids, emb, loss = ops.gumbel_forward(x, cb, U, float(g["temperature"]), float(g["beta"]))
    gx, gc = ops.gumbel_backward(x, cb, U, float(g["temperature"]), float(g["beta"]), g_emb=torch.from_numpy(g["g_emb"]).cuda(), g_loss=torch.from_numpy(g["g_loss"]).cuda())
    rel = lambda a, b: float(np.abs(a.cpu().numpy() - b).max() / np.abs(b).max())
    print(f"  {name}: max err / max |ref|: emb {rel(emb, g['embeddings']):.2e}  loss {rel(loss, g['loss']):.2e}  grad_x {rel(gx, g['grad_x']):.2e}  grad_codebook {rel(gc, g['grad_codebook']):.2e}")
B, D, K, T = 20000, 32, 256, 0.2
gen = torch.Generator().manual_seed(5)
x = torch.randn(B, D, generator=gen) * 0.5; cb = torch.randn(K, D, generator=gen) * 0.3; U = torch.rand(B, K, generator=gen)
xd, cd, Ud = x.double(), cb.double(), U.double()
dist = (xd ** 2).sum(1, keepdim=True) + (cd ** 2).sum(1)[None] - 2 * xd @ cd.T
w = torch.softmax((-dist - torch.log(-torch.log(Ud + 1e-20) + 1e-20)) / T, dim=-1)
emb64 = w @ cd; loss64 = 1.25 * ((xd - emb64) ** 2).sum(1)
ids, emb, loss = ops.gumbel_forward(x.cuda(), cb.cuda(), U.cuda(), T, 0.25)
print(f"  20 000 rows vs fp64: emb max abs err {float((emb.cpu().double() - emb64).abs().max()):.2e} (max |emb| {float(emb64.abs().max()):.2f}), "
      f"loss max rel err {float(((loss.cpu().double() - loss64).abs() / loss64).max()):.2e}")
PY
done
