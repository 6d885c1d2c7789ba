#!/usr/bin/env python3
"""CPU study for DESIGN.md section 9, item 0 (no GPU needed): if the distance scan ran on bf16 triple-split operands
(x = hi + mid + lo exactly, 6 of the 9 cross products, fp32 accumulation) and only rows whose two best approximate
distances are closer than a bound tau were re-decided exactly, how large would tau have to be and how many rows would
take the exact path?

Model: products of two bf16 values are exact in fp32; the accumulation of the 6 x D terms is emulated in float32 in
two orders (term-major and d-major) -- the hardware's internal order is unknown, so the worst of the two and an
analytic worst-case bound are both reported.  The "exact" reference is the oracle's fp32 FMA chain over d.
usage: python tools/bf16_split_study.py [rows]"""
import sys

import numpy as np


def bf16_round(a):
    """float32 -> nearest-even bf16, returned as float32"""
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(a):
    hi = bf16_round(a)
    mid = bf16_round((a - hi).astype(np.float32))
    lo = bf16_round((a - hi - mid).astype(np.float32))
    return hi, mid, lo


def fma_chain(x, c):
    """oracle order: acc = fma(x[:, d], c[:, d], acc), d ascending; emulated with exact float64 products"""
    acc = np.zeros((x.shape[0], c.shape[0]), np.float32)
    for d in range(x.shape[1]):
        acc = (x[:, d:d + 1].astype(np.float64) * c[None, :, d].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    D, K = 32, 256
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    c = (rng.standard_normal((K, D)) * 0.3).astype(np.float32)
    xs, cs = split3(x), split3(c)
    assert np.array_equal((xs[0].astype(np.float64) + xs[1] + xs[2]).astype(np.float32), x), "split must be exact"
    exact = fma_chain(x, c)
    terms = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]  # hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid
    # order A: term-major (each term's D products accumulated, terms from small to large added at the end)
    accA = np.zeros((B, K), np.float32)
    for (i, j) in reversed(terms):
        part = np.zeros((B, K), np.float32)
        for d in range(D):
            part = (part.astype(np.float64) + xs[i][:, d:d + 1].astype(np.float64) * cs[j][None, :, d]).astype(np.float32)
        accA = (accA + part).astype(np.float32)
    # order B: d-major, all six terms of a feature before the next feature
    accB = np.zeros((B, K), np.float32)
    for d in range(D):
        for (i, j) in terms:
            accB = (accB.astype(np.float64) + xs[i][:, d:d + 1].astype(np.float64) * cs[j][None, :, d]).astype(np.float32)
    nx = np.linalg.norm(x, axis=1, keepdims=True).astype(np.float64)
    ncode = np.linalg.norm(c, axis=1)[None, :].astype(np.float64)
    scale = nx * ncode
    for name, acc in (("term-major", accA), ("d-major", accB)):
        err = np.abs(acc.astype(np.float64) - exact.astype(np.float64))
        print(f"{name:10s}: max |approx - exact| = {err.max():.3e}   max relative to |x||c| = {(err / scale).max():.3e}   "
              f"mean = {(err / scale).mean():.3e}")
    xsq = (x.astype(np.float64) ** 2).sum(1, keepdims=True)
    csq = (c.astype(np.float64) ** 2).sum(1)[None, :]
    dist = (xsq + csq - 2.0 * exact).astype(np.float64)
    two = np.partition(dist, 1, axis=1)[:, :2]
    gap = two[:, 1] - two[:, 0]
    worst = 2.0 * (6 * D + 8) * 2.0 ** -24  # 2 x (#accumulations + dropped terms) ulps of |x||c|: analytic worst case
    cmax = ncode.max()
    for label, rel in (("analytic worst case", worst), ("10 x measured max", 10 * max((np.abs(a.astype(np.float64) - exact) / scale).max()
                                                                                          for a in (accA, accB)) * 2)):
        tau = 2.0 * rel * nx[:, 0] * cmax  # both candidates may be off by the bound
        frac = float((gap < tau).mean())
        print(f"tau from {label:20s}: relative bound {rel:.2e} -> {100 * frac:6.3f} % of rows need the exact re-decision "
              f"(P(at least one row in a 32-row tile) = {100 * (1 - (1 - frac) ** 32):.1f} %)")
    print(f"median gap between the two nearest codes: {np.median(gap):.4f}; 1st percentile {np.percentile(gap, 1):.5f}")


if __name__ == "__main__":
    main()
