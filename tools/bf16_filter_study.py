#!/usr/bin/env python3
"""CPU study behind the filtered forward scan (csrc/rq_forward.hip, FILT): how far is the three-term bf16-split dot
product xh.ch + xh.cl + xl.ch (fp32 accumulation) from the oracle's fp32 FMA chain, and how many rows would a threshold
T = 2^e |x| max|c| on the top-2 gap send to the exact re-check -- with zero unflagged argmin mismatches required.
Config-2-like data (20 000 x 32 latents, 256 codes near data points).  Companion of tools/bf16_split_study.py (six terms)."""
import numpy as np
rng=np.random.default_rng(0)
B,K,D=20000,256,32
x=(rng.standard_normal((B,D))*0.5).astype(np.float32)
# codebook: data points (k-means-like), level-0-like
cb=x[rng.choice(B,K,replace=False)]+ (rng.standard_normal((K,D))*0.05).astype(np.float32)
cb=cb.astype(np.float32)
def bf16(a):
    u=a.view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))>>16)<<16
    return r.astype(np.uint32).view(np.float32)
xh=bf16(x); xl=bf16((x-xh).astype(np.float32))
ch=bf16(cb); cl=bf16((cb-ch).astype(np.float32))
# oracle-like dot: two parity chains of fma in fp32 (emulated via float64 then round)
def chain(x,c):
    a0=np.zeros((x.shape[0],c.shape[0]),np.float32); a1=np.zeros_like(a0)
    for d in range(D):
        p=(x[:,d:d+1].astype(np.float64)*c[None,:,d].astype(np.float64))
        if d&1: a1=(p+a1.astype(np.float64)).astype(np.float32)
        else: a0=(p+a0.astype(np.float64)).astype(np.float32)
    return (a0+a1).astype(np.float32)
dot=chain(x,cb)
# split dot with fp32 accumulation, order: for each 16-block: hh, then hl, then lh  (like MFMA chain); emulate per-instruction sum in float64 then round to f32 (optimistic) AND a pessimistic sequential fp32 order
def acc_blocks(pairs):
    acc=np.zeros((B,K),np.float32)
    for (a,b) in pairs:
        for s in range(2):
            # features of k-step s for both halves: kk=8s..8s+7 -> d=2kk+h
            idx=[2*(8*s+j)+h for h in (0,1) for j in range(8)]
            part=np.zeros((B,K),np.float32)
            for d in idx:   # sequential fp32 adds (pessimistic)
                part=(part+ (a[:,d:d+1]*b[None,:,d]).astype(np.float32)).astype(np.float32)
            acc=(acc+part).astype(np.float32)
    return acc
dt=acc_blocks([(xh,ch),(xh,cl),(xl,ch)])
xsq=(x.astype(np.float64)**2).sum(1).astype(np.float32); csq=(cb.astype(np.float64)**2).sum(1).astype(np.float32)
scale=np.sqrt(xsq[:,None].astype(np.float64)*csq[None,:])
err=np.abs(dt.astype(np.float64)-dot.astype(np.float64))/scale
print("max rel err of split dot vs oracle chain (units of sqrt(xsq*csq)):", err.max(), "= 2^", np.log2(err.max()), "mean", err.mean())
tt=(xsq[:,None]+csq[None,:]).astype(np.float32)
d_or=(tt-2*dot).astype(np.float32); d_ap=(tt-2*dt).astype(np.float32)
derr=np.abs(d_ap.astype(np.float64)-d_or)/scale
print("max |d~-d|/sqrt(xsq csq):", derr.max(), "2^",np.log2(derr.max()))
ids=d_or.argmin(1); ida=d_ap.argmin(1)
print("argmin mismatches without guard:", (ids!=ida).sum())
s=np.sort(d_ap,axis=1); gap=s[:,1]-s[:,0]
csqmax=csq.max()
for e in (-13,-12.5,-12,-11.5,-11,-10.5):
    T=2.0**e*np.sqrt(xsq*csqmax)
    flagged=gap<=T
    bad=((ids!=ida)&~flagged).sum()
    print(f"T=2^{e} sqrt(xsq csqmax): flagged {flagged.mean()*100:.3f}% rows, unflagged mismatches {bad}")

# ---- second regime: rows much larger than every code (or the reverse) ------------------------------------------------
# All distances of a row share their leading digits; the final rounding of d (half an ulp of ~|x|^2) decides between
# neighbouring codes, so the threshold needs its 2^-20 (|x|^2 + max|c|^2) term.
print("\nrows 1e5 x larger than the codes:")
xb = (rng.standard_normal((B, D)) * 1e5).astype(np.float32)
cbb = rng.standard_normal((K, D)).astype(np.float32)
xhb = bf16(xb); xlb = bf16((xb - xhb).astype(np.float32))
chb = bf16(cbb); clb = bf16((cbb - chb).astype(np.float32))
x, cb, xh, xl, ch, cl = xb, cbb, xhb, xlb, chb, clb      # (acc_blocks / chain read the module-level operands)
dot = chain(x, cb)
dt = acc_blocks([(xh, ch), (xh, cl), (xl, ch)])
xsq = (x.astype(np.float64) ** 2).sum(1).astype(np.float32); csq = (cb.astype(np.float64) ** 2).sum(1).astype(np.float32)
tt = (xsq[:, None] + csq[None, :]).astype(np.float32)
d_or = (tt - 2 * dot).astype(np.float32); d_ap = (tt - 2 * dt).astype(np.float32)
ids, ida = d_or.argmin(1), d_ap.argmin(1)
s = np.sort(d_ap, axis=1); gap = s[:, 1] - s[:, 0]
T1 = 2.0 ** -12 * np.sqrt(xsq * csq.max()); T2 = T1 + 2.0 ** -20 * (xsq + csq.max())
mis = ids != ida
print(f"argmin mismatches without guard: {mis.sum()}; unflagged with the dot-product term alone: {(mis & (gap > T1)).sum()}; "
      f"with both terms: {(mis & (gap > T2)).sum()} (flagged {100 * (gap <= T2).mean():.1f}% of the rows)")
