#!/usr/bin/env python3
"""Round-5 diagnostics run on the GPU box (prints, does not assert): the wide-layer backward, piece by piece."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import linear as lin  # noqa: E402
from rqhip import ops  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


for R in (1536, 2048, 4096):
    g = torch.Generator().manual_seed(R)
    M = 4099
    x = torch.randn(M, R, generator=g).cuda()
    w = (torch.randn(256, R, generator=g) / R ** 0.5).cuda()
    gy = torch.randn(M, 256, generator=g).cuda()
    y = lin.forward(x, w, True, torch.zeros(256, device="cuda"))
    ref_y = torch.relu(x.double() @ w.double().t())
    print(f"R={R}: forward rel err {rel(y, ref_y):.2e}")
    gm_ref = torch.where(ref_y > 0, gy.double(), torch.zeros_like(ref_y))
    r, c, gmask = ops.maxima(gy, y, rows=True, cols=True, write_masked=True)
    print(f"  maxima: masked equal {torch.equal(gmask.double(), torch.where(y > 0, gy, torch.zeros_like(gy)).double())}, "
          f"rows equal {torch.equal(r[0].view(torch.float32), gmask.abs().amax(dim=1))}, cols equal {torch.equal(c.view(torch.float32), gmask.abs().amax(dim=0))}")
    gx_ref = gm_ref @ w.double()
    img = ops.weight_planes(w, transpose=True, arith=ops.F16X2)
    gx1 = ops.gemm_split_ex(gmask, img, R, a_row_max=r)[0]
    print(f"  dgrad direct (maxima rows): rel err {rel(gx1, gx_ref):.2e}")
    gx2 = lin.input_grad(gmask, w, g_scales=lin.Scales(r, c))
    print(f"  lin.input_grad: rel err {rel(gx2, gx_ref):.2e}")
    xc = ops.maxima(x, rows=False)[1]
    print(f"  x col maxima equal {torch.equal(xc.view(torch.float32), x.abs().amax(dim=0))}")
    gw, _ = ops.linear_wgrad(gmask, None, x, g_col_max=c, x_col_max=xc)
    print(f"  wgrad rel err {rel(gw, gm_ref.t() @ x.double()):.2e}")
    gx3 = lin.input_grad(gmask, w, g_scales=lin.Scales(r, c))
    print(f"  lin.input_grad AFTER the wgrad: rel err {rel(gx3, gx_ref):.2e}")
    gx4, gw4 = lin.backward(gy, y, x, w, True, True, None)
    print(f"  lin.backward: gx rel err {rel(gx4, gx_ref):.2e}, gw rel err {rel(gw4, gm_ref.t() @ x.double()):.2e}")
    bad = ((gx4.double() - gx_ref).abs() > 1e-3 * gx_ref.abs().max()).nonzero()
    if bad.shape[0]:
        print(f"    bad elements {bad.shape[0]}: rows {bad[:,0].min().item()}..{bad[:,0].max().item()} cols {bad[:,1].min().item()}..{bad[:,1].max().item()}; "
              f"ratio at first {(gx4[bad[0,0], bad[0,1]] / gx_ref[bad[0,0], bad[0,1]]).item():.3f}")
