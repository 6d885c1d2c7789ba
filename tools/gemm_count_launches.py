#!/usr/bin/env python3
"""Developer tool: the split-GEMM launches of one configuration-2 training step with their epilogue and maxima flags.
Usage (GPU box): python tools/gemm_count_launches.py"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, ROOT+"/rq-vae-recommender_amd"]
import torch, bench
from data.schemas import SeqBatch
from rqhip import ops, dist as rqdist, tuning
from rqhip.optim import FlatAdamW
tuning.enable_tuned_gemms()
dev=torch.device("cuda",0)
X=torch.nn.functional.normalize(torch.randn(100000,768),dim=-1).to(dev)
model,_=bench.build_model(dev,X[:20000],3,256)
red=rqdist.FlatGradReducer(model.parameters()).attach(model)
opt=FlatAdamW(model.parameters(),lr=1e-3,weight_decay=1e-4)
b=SeqBatch(None,None,None,X,None,None)
def step():
    red.zero_(); out=model(b,gumbel_t=0.2); out.loss.backward(); opt.step()
step(); step()
real=ops.gemm_split_ex
log=[]
def cnt(a,image,n_cols,**k):
    log.append((tuple(a.shape),n_cols,k.get("epilogue",0),bool(k.get("want_row_max")),k.get("col_max_out") is not None))
    return real(a,image,n_cols,**k)
ops.gemm_split_ex=cnt
import rqhip.linear as L
L.ops.gemm_split_ex=cnt
step()
for l in log: print(l)
