#!/usr/bin/env python3
"""Developer probe: how fast are the PyTorch-ROCm fp32 GEMMs of the RQ-VAE MLPs (fwd + bwd) under the BLAS
backends / TunableOp, at the bench shapes (B = 100 000)?"""
import os
import sys
import time

import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dims = [768, 512, 256, 128, 32]
torch.manual_seed(0)
dev = "cuda"
x = torch.randn(B, 768, device=dev)


def mlp(ds):
    layers = []
    for i, (a, b) in enumerate(zip(ds[:-1], ds[1:])):
        layers.append(torch.nn.Linear(a, b, bias=False))
        if i != len(ds) - 2:
            layers.append(torch.nn.ReLU())
    return torch.nn.Sequential(*layers).to(dev)


enc, dec = mlp(dims), mlp(dims[::-1])


def step():
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.grad = None
    z = enc(x)
    xh = dec(z)
    ((xh - x) ** 2).sum(-1).mean().backward()


def bench(tag):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    fl = 3 * 2 * 2 * sum(a * b for a, b in zip(dims[:-1], dims[1:])) * B
    print(f"{tag:40s} {ms:8.3f} ms/step  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


torch.set_float32_matmul_precision("high")
for lib in ("default", "hipblaslt", "cublas"):
    try:
        if lib != "default":
            torch.backends.cuda.preferred_blas_library(lib)
        bench(f"blas={lib} precision=high")
    except Exception as e:  # noqa
        print(lib, "failed:", e)
torch.set_float32_matmul_precision("highest")
bench("blas=last precision=highest")
if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1":
    bench("tunableop (after tuning)")
