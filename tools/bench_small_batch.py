#!/usr/bin/env python3
"""The reference's real training shapes are launch-bound on a GPU: eager step time, kernel launches per step, and the
same step replayed from a hipGraph (torch.cuda.CUDAGraph).
  python tools/bench_small_batch.py            # rqvae_amazon.gin: batch 640, D = 32, STE
  python tools/bench_small_batch.py c3         # rqvae_ml32m.gin (BASELINE config 3): batch 64, D = 64, rotation trick, lr 1e-4
  python tools/bench_small_batch.py --no-jobs  # A/B: the per-layer weight-gradient kernels instead of the job table (csrc/wgrad_jobs.hip)
  python tools/bench_small_batch.py --no-small # A/B: the library GEMMs + mask launches instead of the layers' own kernels (csrc/mlp_small.hip)
  python tools/bench_small_batch.py --json     # both, one JSON line on stdout (bench.py's `secondary.small_batch` runs this in a
                                               # subprocess with a time limit: a graph replay that hangs cannot take the bench line with it)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from rqhip import tuning  # noqa: E402

tuning.enable_tuned_gemms()
torch.autograd.set_multithreading_enabled(False)     # as train_rqvae.train: one GPU per process, the backward on the calling thread
JSON = "--json" in sys.argv
if "--no-jobs" in sys.argv:      # A/B: round 4's per-layer weight-gradient kernels instead of csrc/wgrad_jobs.hip
    from rqhip import linear as _linear
    _linear.use_wgrad_jobs(False)
if "--no-seam" in sys.argv:      # A/B: round 5's library GEMMs for the 128 <-> 32 layers instead of the seam kernel (rqhip_rq_seam)
    from rqhip import linear as _linear
    _linear.use_chain_gemms(False)
if "--no-small" in sys.argv:     # A/B: round 5's library GEMMs + threshold_backward launches instead of csrc/mlp_small.hip
    from rqhip import linear as _linear
    _linear.use_small_kernels(False)
if "--no-cross-stack" in sys.argv:   # A/B: one job-table weight-gradient launch per MLP stack instead of one for both (rqhip/linear.py:xsmall_*)
    from rqhip import linear as _linear
    _linear.use_wgrad_cross_stack(False)
ARGS = [a for a in sys.argv[1:] if a not in ("--json", "--no-jobs", "--no-seam", "--no-small", "--no-cross-stack")]


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def run(c3: bool, B: int, say=print):
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=64 if c3 else 32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False,
              codebook_mode=QuantizeForwardMode.ROTATION_TRICK if c3 else QuantizeForwardMode.STE).cuda()
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.05 / (l + 1)))
    from rqhip.optim import FlatAdamW
    opt = FlatAdamW(m.parameters(), lr=1e-4 if c3 else 1e-3, weight_decay=0.01 if c3 else 1e-4)     # as train_rqvae.py
    x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
    batch = SeqBatch(None, None, None, x, None, None)

    # the step as train_rqvae.train runs it (_GraphedStep._step): gradients in one flat buffer (rqhip.dist.FlatGradReducer, zeroed per step), a
    # cached seed for backward()
    from rqhip import dist as rqdist
    red = rqdist.FlatGradReducer(m.parameters()).attach(m)
    seed = torch.ones((), dtype=torch.float32, device="cuda")

    def step():
        red.zero_()
        out = m(batch, 0.2)
        out.loss.backward(gradient=seed)
        opt.step()
        return out.loss

    res = {"batch": B, "embed_dim": 64 if c3 else 32, "mode": "rotation" if c3 else "ste",
           "gin": "configs/rqvae_ml32m.gin" if c3 else "configs/rqvae_amazon.gin"}
    ms = timeit(step)
    res["eager_ms"], res["eager_items_per_s"] = round(ms, 4), round(B / ms * 1e3, 1)
    say(f"eager  B={B}: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} items/s")
    try:   # kernel launches of one eager step
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        n = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
        ours = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "rqhip" in e.name)
        res["launches_per_step"], res["hand_written_launches"] = n, ours
        say(f"launches per step: {n} device activities ({ours} hand-written kernels)")
        if os.environ.get("RQ_LIST_LAUNCHES"):
            for e in prof.events():
                if e.device_type == torch.autograd.DeviceType.CUDA:
                    say(f"    {e.name[:110]:110s} {e.device_time:8.1f} us")
    except Exception as e:  # noqa
        say("launch count unavailable: " + repr(e)[:200])
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step()
        ms = timeit(g.replay)
        res["graph_ms"], res["graph_items_per_s"] = round(ms, 4), round(B / ms * 1e3, 1)
        say(f"graph  B={B}: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} items/s   (loss {float(loss):.5f})")
    except Exception as e:  # noqa
        res["graph_error"] = repr(e)[:300]
        say("graph capture failed: " + repr(e)[:500])
    return res


if JSON:
    out = {}
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)                       # library chatter -> stderr; the JSON line alone on stdout
    log = lambda *a: print(*a, file=sys.stderr)   # noqa: E731
    out["batch640_amazon"] = run(False, 640, log)
    out["batch64_ml32m"] = run(True, 64, log)
    # the same two steps on round 5's library GEMMs + mask launches (rqhip.linear.use_small_kernels(False)): the A/B of csrc/mlp_small.hip
    from rqhip import linear as _linear
    was = _linear.use_small_kernels(False)
    try:
        for key, args in (("batch640_amazon", (False, 640)), ("batch64_ml32m", (True, 64))):
            r = run(*args, log)
            out[key]["library_gemms"] = {k: r[k] for k in ("eager_ms", "graph_ms", "launches_per_step", "hand_written_launches") if k in r}
    finally:
        _linear.use_small_kernels(was)
    try:     # the gin-driven training LOOP at the reference's corpus size: both step shapes of an epoch replayed vs round 5's form vs eager
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import epoch_tail_ab
        out["train_loop_amazon"] = epoch_tail_ab.measure(1500, log)
    except Exception as e:  # noqa: BLE001
        out["train_loop_amazon"] = {"error": repr(e)[:300]}
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()
else:
    C3 = len(ARGS) > 0 and ARGS[0] == "c3"
    run(C3, 64 if C3 else (int(ARGS[0]) if ARGS else 640))
