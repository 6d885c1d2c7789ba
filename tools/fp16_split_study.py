#!/usr/bin/env python3
"""CPU study for DESIGN section 9 item 0(a): could the activation GEMMs use TWO fp16 pieces per fp32 operand (products
hh + hm + mh: 3 matrix instructions instead of the 6 of the three-piece bf16 split) and still be no less exact than the
library's fp32 GEMM?  Every piece product of two fp16 values is exact in fp32 (11 + 11 bits), accumulation is fp32 -- what
decides is the representation error of the split (22 bits at best) and fp16's narrow exponent range (subnormals below 6.1e-5).
Emulated with torch on the CPU: pieces are formed exactly as a kernel would (round to nearest fp16 / bf16), each piece GEMM is an
fp32 matmul of exactly representable values, the piece GEMMs are added smallest first.
Operands: (1) the model's own shapes -- unit-norm 768-d rows against randn / sqrt(K) weights, post-ReLU hidden activations;
(2) rows and weight rows scaled over twelve decades (tests/test_gpu_gemm_split.py's second gate), with and without a power-of-two
normalisation of every row of A and of B (exact, undone in the epilogue).
usage: python tools/fp16_split_study.py"""
import torch

torch.manual_seed(0)
torch.set_num_threads(8)


def bf16_pieces(v, n):
    out, r = [], v.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def fp16_pieces(v, n, low_shift=0):
    """v = h + m (+ l); pieces after the first are stored multiplied by 2^low_shift (exactly) to stay out of fp16's subnormals"""
    out, r = [], v.clone()
    for i in range(n):
        s = 2.0 ** (low_shift * i)
        p = (r * s).to(torch.float16).to(torch.float32) / s
        out.append(p)
        r = r - p
    return out


def mm(a, b):   # fp32 GEMM of exactly representable pieces (what the matrix instruction accumulates)
    return a @ b.t()


def row_pow2(x):   # exact power-of-two scale per row: largest |value| of the row in [1, 2)
    m = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-38)
    return torch.exp2(torch.floor(torch.log2(m)))


def study(name, A, B):
    ref = A.double() @ B.double().t()
    scale = ref.abs().max().item()
    bound = (A.double().abs() @ B.double().abs().t()) * ((A.shape[1] ** 0.5 + 8) * 2.0 ** -24) + 1e-300
    res = {}

    def rec(tag, C):
        e = (C.double() - ref).abs()
        res[tag] = (e.max().item() / scale, (e / bound).max().item())

    rec("library fp32", A @ B.t())
    a3, b3 = bf16_pieces(A, 3), bf16_pieces(B, 3)
    rec("bf16 x3, 6 products (ships)", ((((mm(a3[1], b3[1]) + mm(a3[2], b3[0])) + mm(a3[0], b3[2])) + mm(a3[1], b3[0])) + mm(a3[0], b3[1])) + mm(a3[0], b3[0]))
    for norm in (False, True):
        sa, sb = (row_pow2(A), row_pow2(B)) if norm else (torch.ones(A.shape[0], 1), torch.ones(B.shape[0], 1))
        An, Bn = A / sa, B / sb
        for shift in (0, 8, 11):
            a2, b2 = fp16_pieces(An, 2, shift), fp16_pieces(Bn, 2, shift)
            c3 = ((mm(a2[1], b2[0]) + mm(a2[0], b2[1])) + mm(a2[0], b2[0])) * sa * sb.t()
            c4 = (((mm(a2[1], b2[1]) + mm(a2[1], b2[0])) + mm(a2[0], b2[1])) + mm(a2[0], b2[0])) * sa * sb.t()
            tag = f"fp16 x2 {'row-normalised' if norm else 'raw'}, low piece x 2^{shift}"
            rec(tag + ", 3 products", c3)
            rec(tag + ", 4 products", c4)
    # block fixed point: every row scaled by an exact power of two to |value| < 1/2, rounded to 23 fractional bits, cut into three
    # signed 8-bit limbs (balanced digits); limb products are exact in int32 (v_mfma_i32_32x32x32_i8 runs at twice the bf16
    # rate), the six limb pairs of weight >= 2^-32 are kept (as the bf16 split keeps six piece pairs)
    def limbs(x):
        s = row_pow2(x) * 4.0                                     # |x| < s / 2: the top limb stays within a signed byte
        q = torch.round((x / s).double() * 2.0 ** 23)            # |q| <= 2^23
        out = []
        for _ in range(3):
            d = torch.remainder(q + 128, 256) - 128               # balanced digit in [-128, 127]
            out.append(d)
            q = (q - d) / 256
        return out, s                                              # x ~ s 2^-23 (d0 + 256 d1 + 65536 d2)
    (a0, a1, a2), sa = limbs(A)
    (b0, b1, b2), sb = limbs(B)
    acc = (a2 @ b2.t()) * 2.0 ** 32 + (a2 @ b1.t() + a1 @ b2.t()) * 2.0 ** 24 + (a2 @ b0.t() + a0 @ b2.t() + a1 @ b1.t()) * 2.0 ** 16
    rec("int8 x3 limbs per row exponent, 6 limb products", (acc * 2.0 ** -46 * sa.double() * sb.double().t()).float())
    print(f"== {name}: A {tuple(A.shape)}, B {tuple(B.shape)}   (max err / max|C|,  max err / |A||B| bound of the test)")
    for k, (e, b) in res.items():
        print(f"   {k:58s} {e:10.3e}   {b:8.3f}")


M, K, N = 4096, 768, 512
x = torch.nn.functional.normalize(torch.randn(M, K), dim=-1)
w = torch.randn(N, K) / K ** 0.5
study("encoder layer 1 (unit-norm rows)", x, w)
h = torch.relu(x @ w.t())
w2 = torch.randn(256, N) / N ** 0.5
study("encoder layer 2 (post-ReLU activations)", h, w2)
g = torch.randn(M, N) * 1e-5 * (torch.rand(M, N) > 0.5)      # a masked, small data gradient
study("data gradient (1e-5-scale, half masked)", g, w.t().contiguous())
a = torch.randn(M, K) * torch.pow(10.0, torch.randint(-6, 7, (M, 1)).float())
b = torch.randn(N, K) * torch.pow(10.0, torch.randint(-3, 4, (N, 1)).float())
study("twelve decades of row scales", a, b)
c = torch.randn(M, K) * torch.pow(10.0, torch.randint(-4, 1, (M, K)).float())   # wide dynamic range INSIDE a row
study("five decades inside every row", c, w)


# ---- the weight gradient dW = g^T x: the reduction runs over the rows, so the power-of-two scale has to be per COLUMN -----------
def study_wgrad(name, G, X):
    ref = G.double().t() @ X.double()
    scale = ref.abs().max().item()
    res = {}

    def rec(tag, Cm):
        res[tag] = (Cm.double() - ref).abs().max().item() / scale

    rec("library fp32", G.t() @ X)
    g3, x3 = bf16_pieces(G, 3), bf16_pieces(X, 3)
    t = lambda a, b: a.t() @ b   # noqa: E731
    rec("bf16 x3, 6 products (ships)", ((((t(g3[1], x3[1]) + t(g3[2], x3[0])) + t(g3[0], x3[2])) + t(g3[1], x3[0])) + t(g3[0], x3[1])) + t(g3[0], x3[0]))
    for norm in (False, True):
        sg = row_pow2(G.t().contiguous()).t() if norm else torch.ones(1, G.shape[1])     # [1, N]: per column of G
        sx = row_pow2(X.t().contiguous()).t() if norm else torch.ones(1, X.shape[1])
        g2, x2 = fp16_pieces(G / sg, 2), fp16_pieces(X / sx, 2)
        rec(f"fp16 x2 {'column-normalised' if norm else 'raw'}, 3 products", ((t(g2[1], x2[0]) + t(g2[0], x2[1])) + t(g2[0], x2[0])) * sg.t() * sx)
    print(f"== wgrad {name}: G {tuple(G.shape)}, X {tuple(X.shape)}   (max err / max|dW|)")
    for k, e in res.items():
        print(f"   {k:58s} {e:10.3e}")


Mw = 32768
xw = torch.nn.functional.normalize(torch.randn(Mw, 768), dim=-1)
hw = torch.relu(xw @ (torch.randn(512, 768) / 768 ** 0.5).t())
gw = torch.randn(Mw, 512) * (1.0 / Mw) * (hw > 0)                                   # a masked gradient at the 1 / B scale
study_wgrad("layer 1 (masked 1/B-scale gradient x unit-norm input)", gw, xw)
gh = torch.randn(Mw, 256) * (1.0 / Mw) * torch.pow(10.0, torch.randint(-3, 1, (Mw, 1)).float())   # rows spread over three decades
study_wgrad("layer 2 (gradient rows over three decades x post-ReLU activations)", gh, hw)
