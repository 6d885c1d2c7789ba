#!/usr/bin/env python3
"""Developer probe (GPU box): one Gumbel-softmax level forward / backward (rqhip_gumbel_forward / _backward) at
three batch sizes, next to the cost of drawing the uniform noise U with torch.rand."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rq-vae-recommender_amd"))
import torch
from rqhip import ops
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e6
for B,D,K in ((100000,32,256),(8192,32,256),(640,32,256)):
    g=torch.Generator(device="cuda").manual_seed(0)
    x=torch.randn(B,D,device="cuda",generator=g)*0.5; cb=torch.randn(K,D,device="cuda",generator=g)*0.3
    U=torch.rand(B,K,device="cuda",generator=g)
    f=timed(lambda: ops.gumbel_forward(x,cb,U,0.2,0.25))
    ids,emb,loss=ops.gumbel_forward(x,cb,U,0.2,0.25)
    ge=torch.randn(B,D,device="cuda",generator=g); gl=torch.full((B,),1.0/B,device="cuda")
    b=timed(lambda: ops.gumbel_backward(x,cb,U,0.2,0.25,g_emb=ge,g_loss=gl))
    r=timed(lambda: torch.rand(B,K,device="cuda"))
    print(f"gumbel level B={B} D={D} K={K}: fwd {f:8.1f} us  bwd {b:8.1f} us  (torch.rand for U: {r:7.1f} us; U is {B*K*4/1e6:.0f} MB)")
