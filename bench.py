#!/usr/bin/env python3
"""bench.py -- items quantised per second through the RQ-VAE hot path on MI355X.

Workloads (BASELINE.json `configs`):
  --config c2 (default; configs[1], the configuration the metric is quoted on at one GPU): synthetic
      100 000 x 768 unit-norm item embeddings per GPU, RQ-VAE 768 -> [512,256,128] -> 32, 3 x 256 codebooks,
      STE forward mode (Gumbel off), beta 0.25, fp32.  One STEP = one pass of the hot path over one 100 000-row
      batch already resident in HBM.
  --config c4 (configs[3]): 10 M x 768 items over 8 GPUs = 1 250 000 rows per GPU (seed 1234 + rank), 4 x 1024
      codebooks, D = 32.  One STEP = one pass over the rank's whole shard as 10 micro-batches of 125 000 rows with
      gradient accumulation, then ONE all-reduce and ONE AdamW update.
A pass = RqVae.forward (encoder GEMMs, the seam launch -- last encoder Linear + every quantisation level + first decoder Linear --,
decoder GEMMs with the reconstruction loss in the last one's epilogue, loss means, id statistics) + backward (HIP closed-form RQ
backward, split-fp16 data-gradient GEMMs and weight-gradient kernels) + one flat-buffer
gradient all-reduce over RCCL (N > 1) + AdamW.  `value` = rows processed by all ranks / wall time of K steps (max over
ranks, barrier + synchronize on both sides).  Weak scaling: every rank owns its own shard.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL).

Extra objects on the JSON line:
  roofline     -- the dominant kernel FAMILY of the step: the MLP matrix kernels (gemm_f16_kernel / gemm_split_kernel activation GEMMs with
                  their epilogues + wgrad_split_kernel weight gradients, ~90 % of the step).  `achieved` = ISSUED matrix-instruction TFLOP/s
                  (three fp16 piece products per fp32 product) over the time of the family's launches, by HIP events recorded on the launch
                  stream inside the first steps of the timed region (rqhip_profile_*); `peak` = 2 500 TFLOP/s dense fp16; the algorithmic
                  (fp32-equivalent) figure beside it.  `traffic` = mean HBM bytes per launch over the family, `traffic_ratio` = PMC bytes /
                  algorithmic bytes (and time-weighted), from separate rocprofv3 --pmc passes matched to launch shapes
                  (tools/profile_bench.sh, profiles/r0N_pmc_kernels_<cfg>.json), used only when that file carries the sha256 of the
                  librqhip.so loaded here.
  roofline_rq  -- the quantisation launch of the step: rq_seam_kernel (128 -> 32 GEMM + all levels + 32 -> 128 GEMM + ReLU in one launch)
                  where the step takes it, else rq_forward_kernel; algorithmic fp32 FLOPs / launch duration / 157.3 TFLOP/s dense fp32 MFMA.
                  `all_fp32_kernel` = rq_forward with RQHIP_FWD_SCAN_FP32, timed in this run (untimed region).
  box          -- two fixed library workloads timed after the step (bf16 8192^3 GEMM TFLOP/s, 1 GiB copy GB/s): which kind of box of
                  the pool this line was measured on (the same build spreads 33-38 M items/s across boxes).
  long_run     -- when the K timed steps took less than --min-seconds (default 1 s; the driver's K = 20 is 0.1 s), a
                  second, longer timed region of the same step (same barriers) and its items/s; `value` stays the K-step
                  figure the contract asks for.
  parity       -- untimed gate against reference-generated ids (tests/golden/parity_<config>.npz): exact-match rate
                  of the HIP ids vs the reference on every fixture row, end to end from the 768-d items and at kernel
                  level, with the tie policy of tests/parity_gate.py (every mismatch must be a flagged near-tie).
  cpu_baseline -- the same training step as a torch-CPU port of the reference's tensor program
                  (oracle/torch_port.py) on this box's host cores, best over thread counts, bounded sample; rank 0,
                  N = 1 only.  The reference's own modules timed on the build container: BASELINE.md section 2.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "rq-vae-recommender_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

INPUT_DIM, HIDDEN, EMBED, BETA = 768, [512, 256, 128], 32, 0.25
CONFIGS = {
    # name: (levels, codes, rows per rank per step, micro-batch rows, fixture tag)
    "c2": dict(levels=3, codes=256, rows=100_000, micro=100_000),
    "c4": dict(levels=4, codes=1024, rows=1_250_000, micro=125_000),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA == fp32 vector peak
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 (v_mfma_f32_32x32x16_bf16), same guide


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _relaunch_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` (N > 1) outside torchrun: become `torch.distributed.run` with N local ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="rehearsal of the multi-GPU path on ONE GPU: N ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on "
                         "one device) -- the same torchrun relaunch, init_from_env, armed early all-reduce, long_run and JSON assembly "
                         "as `--gpus N`; the line says `dry_ranks` and its `value` is not a scaling number")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for c2, 16 for c4: > 1 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="rows per rank per step (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-input-scales", action="store_true", help="A/B: run the maxima pass over the input batch inside every step (round 4) instead of "
                                                                   "taking the rows' maxima from the resident corpus (data/processed.py)")
    ap.add_argument("--torch-adamw", action="store_true", help="A/B: torch.optim.AdamW(fused=True) instead of rqhip.optim.FlatAdamW")
    ap.add_argument("--no-strict", action="store_true", help="skip secondary.strict_fp32 (the same step on library fp32 GEMMs, ~100 steps): profiling runs")
    ap.add_argument("--no-small-batch", action="store_true", help="skip secondary.small_batch (a subprocess: batch 640 / batch 64 steps, eager and hipGraph)")
    ap.add_argument("--cpu-rows", type=int, default=8192)
    ap.add_argument("--pmc-file", default=None, help="JSON of separate rocprofv3 --pmc passes (tools/summarize_profile.py)")
    ap.add_argument("--pmc-window", default=None,
                    help="profiling runs (tools/profile_bench.sh): bracket ONE extra step between two marker launches (rqhip_maxima on a 1 x 4 "
                         "matrix: the only maxima_kernel dispatches with a grid of one workgroup) and write the ordered (kind, flops, bytes) tags "
                         "of its matrix-kernel launches to this file -- tools/summarize_profile.py matches the rocprofv3 --pmc dispatches "
                         "between the markers to the tags (profiles/r0N_pmc_kernels_<cfg>.json: HBM bytes per launch SHAPE)")
    ap.add_argument("--mlp", choices=("split", "split6", "library"), default="split",
                    help="A/B only.  `split` (the product): the MLP GEMMs and weight gradients as two fp16 pieces per operand "
                         "under exact power-of-two scales, three piece products (csrc/gemm_split.hip, wgrad_split.hip); `split6`: "
                         "round 3's three bf16 pieces and six products; `library`: library fp32 GEMMs and the fp32-MFMA weight "
                         "gradients (round 2's step)")
    ap.add_argument("--wide-tiles", action="store_true", help="A/B (round 6, slower): the 512-column layers on 128 x 512 tiles of 8 waves, one workgroup "
                                                              "per CU, instead of 128 x 256 tiles of two workgroups per CU (profiles/r06_gemm_wide_ab.txt)")
    ap.add_argument("--no-cross-stack", action="store_true", help="A/B: the decoder's batched weight gradients launched with the decoder's backward instead of "
                    "waiting for the encoder's launch")
    ap.add_argument("--no-wgrad-batch", action="store_true", help="A/B: one weight-gradient launch per layer (rounds 3-5) instead of one per MLP stack for the "
                    "layers tiled 256 x 256")
    ap.add_argument("--dispenser", action="store_true", help="A/B: the product GEMM's tiles from the atomic dispenser of rounds 4-5 instead of the static schedule")
    ap.add_argument("--tiny-tiles", action="store_true", help="A/B (neutral): the product GEMM's leftover tiles 32 rows high where the launch plan prices them cheaper")
    ap.add_argument("--no-seam", action="store_true", help="A/B: the 128 <-> 32 layers either side of the quantiser as in round 5 (library / split GEMMs, "
                                                           "separate maxima and mask passes) instead of the seam kernel (rqhip_rq_seam: fused forward launch, "
                                                           "its GEMMs as the data gradients)")
    ap.add_argument("--no-trims", action="store_true", help="A/B: without round 6's launch-overhead trims (pooled zero arenas, duplicate statistic on a side "
                                                            "stream, many-workgroup loss means)")
    ap.add_argument("--lib", default=None, help="developer A/B: another build of librqhip.so (same ABI) instead of the in-tree one")
    ap.add_argument("--no-narrow", action="store_true", help="A/B: layers of 128 (mod 256) columns on the library instead of the split kernel's 128-column tile")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="if the K timed steps took less, also time a longer region and report it as `long_run`")
    return ap.parse_args()


def build_model(device, x_init, levels, codes):
    """Weights from torch.manual_seed(0) construction; codebooks by the HIP k-means on the first 20 000 rows
    (np / torch seeds fixed), the reference's own warm-up (train_rqvae.py:178-183)."""
    import numpy as np
    import torch
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    np.random.seed(0)
    model = RqVae(input_dim=INPUT_DIM, embed_dim=EMBED, hidden_dims=HIDDEN, codebook_size=codes, n_layers=levels,
                  n_cat_features=0, codebook_kmeans_init=True, codebook_mode=QuantizeForwardMode.STE,
                  commitment_weight=BETA).to(device)
    model.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model(SeqBatch(None, None, None, x_init, None, None), 0.2)   # lazy k-means init of every level
    torch.cuda.synchronize()
    return model, time.perf_counter() - t0


def _cpu_limits():
    """What this process may use of the host: cgroup CPU quota (cores' worth; None = unlimited / unknown) and the affinity mask."""
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                quota = None if q <= 0 else round(q / per, 2)
        except (OSError, ValueError):
            pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    return quota, affinity


def cpu_baseline(batch_rows, levels, codes, budget_s=24.0):
    """Best items/s of the torch-CPU port over thread counts; ~budget_s seconds of CPU work in total."""
    import torch
    from oracle import torch_port
    g = torch.Generator().manual_seed(1234)
    kw = dict(hidden=HIDDEN, embed_dim=EMBED, n_levels=levels, codebook_size=codes, beta=BETA)
    cores = os.cpu_count() or 1
    cand = sorted({t for t in (8, 16, 32, 64) if t <= cores}) or [cores]   # (every hardware thread loses by 300x: not swept)
    x = torch.nn.functional.normalize(torch.randn(batch_rows, INPUT_DIM, generator=g), dim=-1)
    sweep, best = {}, None
    per = budget_s / (len(cand) + 1)
    for t in cand:
        torch.set_num_threads(t)
        probe = torch_port.time_training_steps(x, steps=1, warmup=1, **kw)
        steps = max(2, min(30, int(0.6 * per / max(probe["seconds"], 1e-3))))
        r = torch_port.time_training_steps(x, steps=steps, warmup=0, **kw)
        sweep[str(t)] = round(r["items_per_s"], 1)
        if best is None or r["items_per_s"] > best[1]["items_per_s"]:
            best = (t, r)
    t, r = best
    torch.set_num_threads(t)
    x640 = x[:640].contiguous()
    r640 = torch_port.time_training_steps(x640, steps=max(5, int(per / 0.05)) if per < 3 else 60, warmup=2, **kw)
    torch.set_num_threads(cores)
    quota, affinity = _cpu_limits()
    return {"value": round(r["items_per_s"], 1), "unit": "items/s", "cores": t, "host_hardware_threads": cores, "kind": "port",
            "cgroup_cpu_quota_cores": quota, "affinity_cpus": affinity,
            "sample": f"{r['steps']} fwd+bwd+AdamW steps of {batch_rows} rows ({r['seconds']:.1f} s) at the best of "
                      f"{cand} threads (`cores` = the thread count that won); torch-CPU port of the reference program "
                      f"(oracle/torch_port.py), same model shape",
            "threads_sweep_items_per_s": sweep,
            "batch640_items_per_s": round(r640["items_per_s"], 1),
            "note": "the reference's own modules on the 8-vCPU build container, and this port beside them on that host: "
                    "BASELINE.md section 2, profiles/r04_reference_vs_port_cpu.json (tools/time_reference_cpu.py); "
                    "cgroup_cpu_quota_cores / affinity_cpus: what the box lets this process use of its hardware threads"}


def box_calibration(device):
    """Two fixed library workloads, timed after the step (untimed region): the same build measured 33-38 M items/s on different boxes
    of the pool with EVERY kernel 12-18 % apart (profiles/r05_bench_n1_sample_*.json) -- these two numbers say which kind of box a
    line comes from.  bf16 8192^3 GEMM through the library (matrix pipe under the power cap), 1 GiB device copy (HBM)."""
    try:
        import torch
        a = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
        for _ in range(3):
            a @ b
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            a @ b
        ev[1].record()
        torch.cuda.synchronize()
        gemm = 20 * 2 * 8192 ** 3 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12
        del a, b
        src = torch.empty(1 << 28, device=device, dtype=torch.float32)
        dst = torch.empty_like(src)
        dst.copy_(src)
        ev[0].record()
        for _ in range(10):
            dst.copy_(src)
        ev[1].record()
        torch.cuda.synchronize()
        copy = 10 * 2 * src.numel() * 4 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
        del src, dst
        torch.cuda.empty_cache()
        return {"bf16_gemm_8192_tflops": round(gemm, 1), "copy_read_plus_write_GBps": round(copy, 1)}
    except Exception as e:  # noqa
        return {"error": repr(e)[:200]}


def small_batch_secondary(cpu):
    """The batch sizes the reference's gin files ship (amazon: 640, D = 32, STE; ml32m: 64, D = 64, rotation trick), eager and
    replayed from a hipGraph: tools/bench_small_batch.py --json in a SUBPROCESS with a time limit."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_small_batch.py"), "--json"]
    try:
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        out = json.loads(pr.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}
    c640 = (cpu or {}).get("batch640_items_per_s")
    if c640:
        a = out.get("batch640_amazon", {})
        out["batch640_vs_cpu_port"] = {"cpu_port_items_per_s": c640,
                                       "eager_ratio": round(a.get("eager_items_per_s", 0.0) / c640, 2),
                                       "graph_ratio": round(a.get("graph_items_per_s", 0.0) / c640, 2) if "graph_items_per_s" in a else None}
    out["note"] = ("one fwd+bwd+AdamW step at the reference's shipped batch sizes (every layer on csrc/mlp_small.hip; `library_gemms` = the "
                   "same step on round 5's library GEMMs + mask launches); `graph` = the same step captured once and replayed "
                   "(train_rqvae.py does this by default below 4096 rows, for the full batch and for the short batch that ends an epoch: "
                   "`train_loop_amazon` = iterations/s of that gin-driven loop on a 12 101-item synthetic corpus, both shapes replayed vs round 5's "
                   "eager tail + re-capture vs eager); cpu port at batch 640: cpu_baseline.batch640_items_per_s")
    return out


def parity_gate(device, tag):
    """HIP ids vs the reference's on every fixture row (untimed).  Returns the `parity` object of the bench line."""
    import numpy as np
    import torch
    from rqhip import ops
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_gate as parity
    fx = parity.load_fixture(tag)
    beta = float(fx["beta"])
    cbs = torch.from_numpy(fx["codebooks"]).to(device)
    out = {"fixture": f"tests/golden/parity_{tag}.npz (reference run by oracle/gen_parity_fixtures.py)",
           "tie_policy": "a row may differ from the reference only where the kernel's tie_margin at the first differing "
                         "level is below tau (tests/parity_gate.py); such rows are adjudicated in fp64"}

    # (1) kernel level, identical input bits: regenerable latents, all rows (the reference's level loop ran on them)
    z = parity.regenerable_latents(int(fx["z_ids_eval"].shape[0]), float(fx["z_scale"]), int(fx["z_seed"]))
    ok_bits = parity.sha(z) == str(fx["z_sha256"])
    k = ops.rq_forward(torch.from_numpy(z).to(device), cbs, ops.MODE_EVAL, beta, want_margin=True, want_embs=False)
    ref = fx["z_ids_eval"].astype(np.int64)
    ids = k.ids.t().cpu().numpy()
    cmp = parity.compare_ids(ids, ref, k.tie_margin.cpu().numpy(), parity.TAU_KERNEL)
    n = len(fx["z_loss_eval_head"])
    keep = np.ones(n, bool)
    keep[cmp["mismatch_rows"][cmp["mismatch_rows"] < n]] = False
    lerr = float(np.abs(k.loss.cpu().numpy()[:n][keep] - fx["z_loss_eval_head"][keep]).max())
    out["kernel_level"] = parity.summary(cmp, {"input": "regenerable latents (bits verified)" if ok_bits else
                                               "regenerable latents (BITS DIFFER FROM FIXTURE)",
                                               "loss_max_abs_err": float(f"{lerr:.3e}")})
    # (1a) the product launch (no margins: for D = 32 the filtered bf16-split scan with exact re-checks) must return the
    # very ids and losses of the all-fp32 margin kernel above, on every row and on the hard rows below
    kp = ops.rq_forward(torch.from_numpy(z).to(device), cbs, ops.MODE_EVAL, beta, want_embs=False, want_residuals=False)
    same = bool(torch.equal(kp.ids, k.ids) and torch.equal(kp.loss, k.loss))
    # (1b) the reference's own encoder-output bits for its 2048 closest calls
    h = ops.rq_forward(torch.from_numpy(fx["hard_res0"]).to(device), cbs, ops.MODE_EVAL, beta, want_margin=True,
                       want_embs=False, want_residuals=False)
    href = parity.reference_ids(fx, False)[fx["hard_rows"]]
    hcmp = parity.compare_ids(h.ids.t().cpu().numpy(), href, h.tie_margin.cpu().numpy(), parity.TAU_KERNEL)
    out["kernel_level_hard_rows"] = parity.summary(hcmp, {"input": "reference res0 bits of the 2048 smallest-margin rows"})
    hp = ops.rq_forward(torch.from_numpy(fx["hard_res0"]).to(device), cbs, ops.MODE_EVAL, beta, want_embs=False,
                        want_residuals=False)
    same = same and bool(torch.equal(hp.ids, h.ids) and torch.equal(hp.loss, h.loss))

    # (2) end to end from the 768-d items through the GPU encoder GEMMs (res0 differs from MKL's in the last bits)
    model = parity.build_fixture_model(fx, device)
    x = parity.synthetic_items(int(fx["n_rows"]), int(fx["x_seed"])).to(device)
    e2e = {}
    worst_loss = 0.0
    for training, mode in ((False, ops.MODE_EVAL), (True, ops.MODE_STE)):
        model.train(training)
        with torch.no_grad():
            res0 = model.encode(x)
            r = ops.rq_forward(res0, cbs, mode, beta, want_margin=True, want_embs=False, want_residuals=False)
            rp = ops.rq_forward(res0, cbs, mode, beta, want_embs=False, want_residuals=False)   # product launch
        same = same and bool(torch.equal(rp.ids, r.ids) and torch.equal(rp.loss, r.loss))
        refi = parity.reference_ids(fx, training)
        c = parity.compare_ids(r.ids.t().cpu().numpy(), refi, r.tie_margin.cpu().numpy(), parity.TAU_E2E)
        p = "train" if training else "eval"
        n = len(fx[f"loss_{p}_head"])
        keep = np.ones(n, bool)
        keep[c["mismatch_rows"][c["mismatch_rows"] < n]] = False
        worst_loss = max(worst_loss, float(np.abs(r.loss.cpu().numpy()[:n][keep] - fx[f"loss_{p}_head"][keep]).max()))
        e2e[p] = parity.summary(c)
    from data.schemas import SeqBatch
    model.train(True)
    with torch.no_grad():
        losses = model(SeqBatch(None, None, None, x, None, None), 0.2)
    e2e["train_step_loss_abs_err"] = float(f"{abs(float(losses.loss) - float(fx['train_loss'])):.3e}")
    e2e["train_step_recon_abs_err"] = float(
        f"{abs(float(losses.reconstruction_loss) - float(fx['train_reconstruction_loss'])):.3e}")
    e2e["quantize_loss_max_abs_err"] = float(f"{worst_loss:.3e}")
    out["end_to_end"] = e2e
    out["product_kernel_equals_margin_kernel"] = same   # ids and losses, bit for bit, kernel level + hard rows + end to end
    # headline fields
    out["ids_exact_rate"] = min(e2e["eval"]["ids_exact_rate"], e2e["train"]["ids_exact_rate"])
    out["mismatches"] = e2e["eval"]["mismatches"] + e2e["train"]["mismatches"]
    out["all_mismatches_flagged"] = bool(e2e["eval"]["all_mismatches_flagged"] and e2e["train"]["all_mismatches_flagged"]
                                         and cmp["all_mismatches_flagged"] and hcmp["all_mismatches_flagged"])
    out["loss_max_abs_err"] = max(worst_loss, lerr, e2e["train_step_loss_abs_err"])
    out["pass"] = bool(out["all_mismatches_flagged"] and out["loss_max_abs_err"] <= 1e-5
                       and cmp["mismatches"] <= cmp["rows_flagged"] and same)
    del model
    return out


def _sha256_file(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    args = parse_args()
    n_ranks = args.dry_ranks if args.dry_ranks > 1 else args.gpus
    if n_ranks > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_under_torchrun(n_ranks)

    import numpy as np
    import torch
    import torch.distributed as dist

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when its communicator is
    # created, so every library's chatter is sent to stderr: fd 1 is parked and only the JSON line goes to it.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from rqhip import dist as rqdist
    from rqhip import autograd as rq_autograd
    from rqhip import ops, tuning
    from data.schemas import SeqBatch

    cfg = CONFIGS[args.config]
    LEVELS, CODES = cfg["levels"], cfg["codes"]
    B = args.batch or cfg["rows"]
    micro = min(cfg["micro"], B)
    steps = args.steps if args.steps is not None else (200 if args.config == "c2" else 16)

    dry = args.dry_ranks > 1
    rank, local_rank, world = (rqdist.init_from_env("cuda", backend="gloo", device_index=0) if dry
                               else rqdist.init_from_env("cuda"))
    if world != n_ranks:
        raise SystemExit(f"--gpus {args.gpus} / --dry-ranks {args.dry_ranks} but WORLD_SIZE={world}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if args.lib:
        from rqhip import _lib as _rqlib
        _rqlib.load(os.path.abspath(args.lib))
    tuned = tuning.enable_tuned_gemms()   # fp32 library-GEMM selections for the MLP layers the split kernels do not tile
    from rqhip import linear as _lin
    _lin.use_arith({"split": "f16x2", "split6": "bf16x3", "library": "fp32"}[args.mlp])   # (A/B arms: tools/profile_mlp_ab.sh)
    _lin.use_narrow_tiles(not args.no_narrow)
    _lin.use_wide_tiles(args.wide_tiles)
    _lin.use_chain_gemms(not args.no_seam)
    _lin.use_tiny_tiles(args.tiny_tiles)
    _lin.use_static_tiles(not args.dispenser)
    _lin.use_wgrad_batch(not args.no_wgrad_batch)
    _lin.use_wgrad_cross_stack(not args.no_cross_stack)
    _lin.use_step_trims(not args.no_trims)
    g = torch.Generator().manual_seed(1234 + rank)
    X = torch.empty((B, INPUT_DIM), device=device)
    for lo in range(0, B, 250_000):        # generated in host chunks: 1.25 M x 768 fp32 is 3.8 GB
        hi = min(B, lo + 250_000)
        X[lo:hi] = torch.nn.functional.normalize(torch.randn(hi - lo, INPUT_DIM, generator=g), dim=-1).to(device)
    model, kmeans_s = build_model(device, X[: min(20000, B)], LEVELS, CODES)
    rqdist.broadcast_module(model)
    from rqhip.optim import FlatAdamW
    opt = (torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True) if args.torch_adamw     # (A/B: torch's fused kernel)
           else FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4))     # the same update, one launch over all parameters (csrc/adamw.hip)
    reducer = rqdist.FlatGradReducer(model.parameters()).attach(model)
    batches = [SeqBatch(None, None, None, X[lo:min(B, lo + micro)], None, None) for lo in range(0, B, micro)]
    n_micro = len(batches)
    # The largest |value| of an item's row is a property of the item: the resident item matrix holds it (data/processed.py computes it once
    # per corpus and hands it out with every batch >= 4096 rows), so the step does not run a maxima pass over its input batch.  Done here
    # once for the resident synthetic corpus X, exactly as ItemData.__getitem__ does.  --no-input-scales: the pass inside every step (round 4).
    if not args.no_input_scales and args.mlp == "split" and micro >= 4096:
        x_rows, x_cols, _ = ops.maxima(X)
        for bi, lo in enumerate(range(0, B, micro)):
            _lin.attach_scales(batches[bi].x, x_rows[:, lo:min(B, lo + micro)].contiguous(), x_cols)

    seed_one = torch.ones((), dtype=torch.float32, device=device)

    def step():
        reducer.zero_()
        for bi, b in enumerate(batches):
            if bi + 1 == n_micro:
                reducer.arm()                   # last backward of the step: finished gradients go on the wire under the encoder's
            share = b.x.shape[0] / B            # this micro-batch's share of the step's mean loss
            with rq_autograd.loss_scale(share):  # (hint for the speculative reconstruction-loss gradient)
                out = model(b, gumbel_t=0.2)
            # (the seed of the backward as a cached tensor: `backward()` without one fills a fresh ones_like(loss) every step -- a launch)
            (out.loss if n_micro == 1 else out.loss * share).backward(gradient=seed_one)
        reducer.allreduce_mean()
        opt.step()
        return out

    for _ in range(args.warmup):
        out = step()
    # the timed region records the kernels that ARE the step -- the MLP GEMM / weight-gradient family -- in its FIRST n_rec steps only:
    # ~21 event pairs per step cost 5 % of a 2.8 ms step (K-step 2.90 ms vs 2.76 ms unrecorded, profiles/r05_bench_n1.json's long_run)
    n_rec = min(steps, 4 if steps >= 100 else 2)     # (short regions -- the driver's K = 20 -- record two steps: the pairs are 1 % of such a region)
    n_fam = n_rec * n_micro * 26 + 64
    ops.profile_enable(n_fam)
    ops.profile_select("gemm_split", "wgrad")
    if world > 1:
        reducer.enable_timing()
    rqdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for si in range(steps):
        if si == n_rec:
            ops.profile_select("none")      # (a host-side flag: no further records, no synchronisation)
        out = step()
    torch.cuda.synchronize()
    rqdist.barrier()
    elapsed = time.perf_counter() - t0
    exposed = reducer.exposed_ms() if world > 1 else []
    reducer.enable_timing(False)
    exposed_all = None
    if world > 1:    # every rank's mean exposed all-reduce time (device events, host clock), gathered on all ranks
        mine = torch.tensor([float(np.mean([d for d, _ in exposed])) if exposed else 0.0,
                             float(np.mean([h for _, h in exposed])) if exposed else 0.0], device=device, dtype=torch.float64)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        exposed_all = [[round(float(p[0]), 4), round(float(p[1]), 4)] for p in parts]
    fam_records = ops.profile_read_tagged(n_fam)
    ops.profile_enable(0)
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # a K-step region shorter than --min-seconds (the driver's K = 20 at 5 ms) is re-measured over a longer one, same
    # protocol; every rank takes the same decision from the reduced time
    long_run = None
    if elapsed < args.min_seconds:
        n_long = int(min(2000, max(steps + 1, round(1.15 * args.min_seconds * steps / elapsed))))
        rqdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_long):
            out = step()
        torch.cuda.synchronize()
        rqdist.barrier()
        e_long = time.perf_counter() - t1
        if dist.is_initialized():
            t = torch.tensor([e_long], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e_long = float(t)
        long_run = {"steps": n_long, "seconds": round(e_long, 4), "ms_per_step": round(e_long / n_long * 1e3, 4),
                    "items_per_s": round(B * world * n_long / e_long, 1),
                    "why": f"the {steps} timed steps took {elapsed:.3f} s < --min-seconds {args.min_seconds}"}

    # ---- untimed from here ------------------------------------------------------------------------------
    if args.pmc_window and rank == 0:
        tiny = torch.zeros((1, 4), device=device)
        ops.maxima(tiny)                                   # marker dispatch
        ops.profile_enable(n_micro * 48 + 64)
        ops.profile_select("gemm_split", "wgrad")
        step()
        torch.cuda.synchronize()
        wrecs = ops.profile_read_tagged(n_micro * 48 + 64)
        ops.profile_enable(0)
        ops.maxima(tiny)                                   # marker dispatch
        torch.cuda.synchronize()
        with open(args.pmc_window, "w") as fh:
            json.dump({"tags": [[k, fl, by] for k, _ms, fl, by in wrecs], "config": args.config, "rows": B, "micro": micro}, fh)
    # Per-kernel records for `roofline_kernels`: the same step, every library launch bracketed by HIP events on its stream
    # (rqhip_profile_*: tag, duration, algorithmic FLOPs and bytes of the launch).  Its own region because ~50 event pairs
    # per step are not free; the records of `roofline` above come from the timed region itself.
    n_prof = min(steps, 20)
    ops.profile_enable(n_prof * n_micro * 48 + 64)
    ops.profile_select()
    torch.cuda.synchronize()
    for _ in range(n_prof):
        out = step()
    torch.cuda.synchronize()
    prof_records = ops.profile_read_tagged(n_prof * n_micro * 48 + 64)
    ops.profile_enable(0)
    kernel_ms = [ms for kind, ms, _fl, _by in prof_records if kind == "rq_forward"]   # the scan kernel (`roofline_rq`)
    # ... or, where RqVae.forward takes the seam node (>= 4096 rows, D = 32 behind a 128-wide hidden layer), the fused launch: the
    # rq_seam records that carry the levels' FLOPs (the bare-GEMM launches of the backward are the same tag with two GEMMs' FLOPs or fewer)
    seam_gemm = 2.0 * micro * EMBED * 128
    seam_recs = [(ms, fl, by) for kind, ms, fl, by in prof_records if kind == "rq_seam" and fl > 2.0 * seam_gemm + 1.0]
    seam_fused = bool(seam_recs) and not kernel_ms
    if seam_fused:
        kernel_ms = [ms for ms, _fl, _by in seam_recs]

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        torch.cuda.synchronize()
        return r, a.elapsed_time(b)

    def reps(fn, n=10):
        fn()
        return timed(lambda: [fn() for _ in range(n)])[1] / n

    # the strict-fp32 arm of the same step (`--mlp library`: library fp32 GEMMs + oracle-ordered fp32-MFMA weight gradients),
    # same process, same model: what the step costs without the f16x2 emulation
    strict = None
    if args.mlp == "split" and world == 1 and not args.no_strict:
        prev_arith = _lin.use_arith("fp32")
        n_strict = max(50, min(steps, 100))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(n_strict):
            step()
        torch.cuda.synchronize()
        e_strict = time.perf_counter() - ts
        _lin.use_arith(prev_arith)
        step()
        torch.cuda.synchronize()
        strict = {"arith": "fp32: library fp32 GEMMs (PyTorch-ROCm, matmul precision highest, TunableOp selections) + fp32-MFMA "
                           "weight gradients in the oracle's order (bench.py --mlp library)",
                  "steps": n_strict, "ms_per_step": round(e_strict / n_strict * 1e3, 4),
                  "items_per_s": round(B * n_strict / e_strict, 1)}

    Xm = batches[0].x
    Bm = Xm.shape[0]
    reducer.zero_()
    res0, enc_ms = timed(lambda: model.encode(Xm))
    out1, fwd_ms = timed(lambda: model(batches[0], gumbel_t=0.2))
    _, bwd_ms = timed(lambda: out1.loss.backward())
    allreduce_ms = reps(reducer.allreduce_mean) if world > 1 else 0.0
    _, opt_ms = timed(lambda: opt.step())

    # secondary scopes of SURVEY.md 8d: S-rq = the quantisation stack alone (HIP forward + HIP backward on the 32-d
    # latents), and tokenisation only (encoder + RQ, eval mode)
    cbs = torch.stack([l.weight for l in model.layers]).detach()
    lat = res0.detach()
    g_sum = torch.randn_like(lat)
    g_l = torch.full((Bm,), 1.0 / Bm, device=device)

    def rq_fwd_bwd():
        o = ops.rq_forward(lat, cbs, 1, BETA, want_embs=False, want_residuals=False)
        ops.rq_backward(lat, cbs, 1, BETA, o.ids, g_embsum=g_sum, g_loss=g_l, cbgrad=ops.cbgrad_default())   # (the training path's form)

    rq_ms = reps(lambda: ops.rq_forward(lat, cbs, 1, BETA, want_embs=False, want_residuals=False))
    # the same launch on the all-fp32 matrix scan, main kernel only (HIP events on the launch stream, as `roofline`)
    ops.profile_enable(64)
    for _ in range(12):
        ops.rq_forward(lat, cbs, 1, BETA, want_embs=False, want_residuals=False, scan="fp32")
    f32_ms = ops.profile_read(64)[2:]
    ops.profile_enable(0)
    srq_ms = reps(rq_fwd_bwd)
    model.eval()
    with torch.no_grad():
        tok_ms = reps(lambda: model.get_semantic_ids(Xm))
    model.train()
    final_loss, p_unique = float(out.loss.detach()), float(out.p_unique_ids)
    reducer_overlaps = reducer.overlap_launches

    # the codebook initialisation alone, warm (the first forward above also pays library / module loading): k-means of
    # every level on its own residuals of the first 20 000 rows, as train_rqvae.py:178-183 triggers it
    from init.kmeans import Kmeans
    with torch.no_grad():
        res_km = model.encode(X[: min(20000, B)])
        np.random.seed(0)
        torch.manual_seed(0)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for _l in range(LEVELS):
            cents = Kmeans(k=CODES).run(res_km).centroids
            res_km = res_km - ops.rq_forward(res_km, cents[None], ops.MODE_EVAL, BETA, want_residuals=False,
                                             want_norm=False).embs[0]
        torch.cuda.synchronize()
        kmeans_only_s = time.perf_counter() - tk

    if rank == 0:
        items = B * world * steps
        value = items / elapsed
        ms_per_step = elapsed / steps * 1e3
        flops_per_row = LEVELS * (2 * EMBED * CODES + 5 * EMBED)            # 49 632 (c2), 262 784 (c4)
        mlp_fwd = 2 * 2 * (INPUT_DIM * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1] * HIDDEN[2] + HIDDEN[2] * EMBED)
        # backward = 2 x forward minus the first layer's input gradient, which is never formed
        step_flops_per_row = 3 * mlp_fwd - 2 * INPUT_DIM * HIDDEN[0] + flops_per_row
        mean_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = flops_per_row * Bm / (mean_ms * 1e-3) / 1e12 if kernel_ms else float("nan")
        bytes_per_row = 8 * EMBED + 12 * LEVELS + 4                          # fwd: 296 B (c2), 308 B (c4)
        rq_flops_row = flops_per_row
        if seam_fused:     # the fused launch: + the two 128 <-> 32 GEMMs; rows of h in, res0 / ids / emb_sum / loss / norms / rows of d out
            rq_flops_row = flops_per_row + 2 * 2 * EMBED * 128
            bytes_per_row = 4 * 128 + 4 * EMBED + 8 * LEVELS + 4 * EMBED + 4 + 4 * LEVELS + 4 * 128
            achieved = rq_flops_row * Bm / (mean_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        from rqhip import _lib as _rqlib
        lib_sha = _sha256_file(os.path.abspath(args.lib) if args.lib else _rqlib.SO_PATH)
        import glob as _glob
        def _newest(pattern):     # profiles/r0N_<pattern>: the newest round first
            return sorted(_glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True)
        pmc = args.pmc_file or next(iter(_newest(f"r[0-9][0-9]_pmc_traffic_{args.config}.json")),
                                    os.path.join(ROOT, "profiles", f"r06_pmc_traffic_{args.config}.json"))
        pmc_json = None
        if not os.path.exists(pmc):
            traffic_src = f"null: no PMC file {os.path.relpath(pmc, ROOT)}"
        elif B != cfg["rows"]:
            traffic_src = "null: the PMC passes were collected at the configuration's batch size, not --batch"
        else:
            with open(pmc) as fh:
                pj = json.load(fh)
            if pj.get("librqhip_sha256") != lib_sha:
                traffic_src = (f"null: {os.path.relpath(pmc, ROOT)} was collected on librqhip.so "
                               f"{str(pj.get('librqhip_sha256'))[:16]}, this run loaded {lib_sha[:16]}")
            else:
                pmc_json = pj
                traffic = pj.get("rq_forward_kernel", {}).get("hbm_bytes_per_launch_corrected")
                traffic_src = (os.path.relpath(pmc, ROOT) + " (2*FETCH_SIZE+WRITE_SIZE, bytes/launch; same librqhip.so "
                               f"sha256 {lib_sha[:16]} as this run)")
        f32_mean = float(np.mean(f32_ms)) if f32_ms else float("nan")
        step_tflops = step_flops_per_row * B / (ms_per_step * 1e-3) / 1e12
        # ---- the kernels that dominate the step ------------------------------------------------------
        issued_mult = {"split": 3.0, "split6": 6.0, "library": 1.0}[args.mlp]   # matrix-instruction FLOPs per algorithmic FLOP
        issued_peak = PEAK_FP32_MFMA_TFLOPS if args.mlp == "library" else PEAK_BF16_MFMA_TFLOPS   # (fp16 and bf16 dense peaks are equal)
        groups = {}
        for kind, ms, fl, by in prof_records:
            key = (kind, fl, by)
            groups.setdefault(key, []).append(ms)
        # HBM bytes per launch SHAPE (tag "kind:flops:bytes"), from the rocprofv3 --pmc passes of tools/profile_bench.sh matched to the tags of
        # a marked step (--pmc-window); used only when collected on the very library this run loaded
        pmc_k, pmc_k_src = {}, "null: no profiles/r0N_pmc_kernels_%s.json for this build (tools/profile_bench.sh)" % args.config
        for pmc_all in _newest(f"r[0-9][0-9]_pmc_kernels_{args.config}.json"):
            with open(pmc_all) as fh:
                pj = json.load(fh)
            if pj.get("librqhip_sha256") == lib_sha and B == cfg["rows"]:
                pmc_k, pmc_k_src = pj.get("kernels", {}), os.path.relpath(pmc_all, ROOT)
                break
        rk = []
        for (kind, fl, by), mss in groups.items():
            mean = float(np.mean(mss))
            matrix = kind in ("gemm_split", "wgrad") and fl >= 2.0 * 4096 * 128 * 256
            e = {"kernel": kind, "calls_per_step": round(len(mss) / n_prof, 2), "mean_us": round(mean * 1e3, 2),
                 "step_us": round(float(np.sum(mss)) / n_prof * 1e3, 1),
                 "algorithmic_gflop": round(fl / 1e9, 3), "algorithmic_mb": round(by / 1e6, 2),
                 "achieved_tflops": round(fl / (mean * 1e-3) / 1e12, 2),
                 "achieved_gbps": round(by / (mean * 1e-3) / 1e9, 1),
                 "frac_vs_fp32_peak": round(fl / (mean * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 "frac_vs_hbm_peak": round(by / (mean * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)}
            if matrix:
                e["issued_tflops"] = round(issued_mult * fl / (mean * 1e-3) / 1e12, 1)
                e["frac_vs_issued_dtype_peak"] = round(issued_mult * fl / (mean * 1e-3) / 1e12 / issued_peak, 4)
            e["bound"] = "mfma" if (matrix or kind == "rq_forward") else "hbm"
            tr = pmc_k.get(f"{kind}:{int(fl)}:{int(by)}")
            e["traffic_ratio"] = round(tr / by, 3) if (tr and by) else None
            rk.append(e)
        rk.sort(key=lambda e: -e["step_us"])
        gemm_us = sum(e["step_us"] for e in rk if "issued_tflops" in e)
        gemm_issued = sum(e["issued_tflops"] * e["step_us"] for e in rk if "issued_tflops" in e) / max(gemm_us, 1e-9)
        roofline_kernels = {
            "how": f"HIP events around every library launch (rqhip_profile_*) over {n_prof} steps run after the timed region; "
                   "FLOPs / bytes are ALGORITHMIC (2 M N K per GEMM; operands and results once); issued = matrix-instruction "
                   f"FLOPs ({issued_mult:g} piece products per product); peaks: fp32 157.3 TF/s, fp16/bf16 dense 2500 TF/s, HBM 8 TB/s",
            "recorded_us_per_step": round(sum(e["step_us"] for e in rk), 1),
            "matrix_kernels": {"us_per_step": round(gemm_us, 1), "share_of_step": round(gemm_us / (ms_per_step * 1e3), 3),
                               "issued_tflops_time_weighted": round(gemm_issued, 1),
                               "frac_of_issued_dtype_peak": round(gemm_issued / issued_peak, 4)},
            "kernels": rk[:12],
        }
        # ---- `roofline`: the dominant kernel FAMILY, from the HIP-event records of the TIMED region ----
        fam = {}
        for kind, ms, fl, by in fam_records:
            if fl >= 2.0 * 4096 * 128 * 256:                  # the matrix kernels proper (the 32-wide layers' fp32 kernels are not)
                fam.setdefault((kind, fl, by), []).append(ms)
        fam_ms = sum(sum(v) for v in fam.values())
        fam_fl = sum(k[1] * len(v) for k, v in fam.items())
        fam_launches = sum(len(v) for v in fam.values())
        fam_alg = fam_fl / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else float("nan")
        dom_key = max(fam, key=lambda k: sum(fam[k])) if fam else None
        dom = None
        if dom_key is not None:
            dk, dfl, dby = dom_key
            dmean = float(np.mean(fam[dom_key]))
            dom = {"kernel": {"gemm_split": "gemm_f16_kernel", "wgrad": "wgrad_split_jobs_kernel / wgrad_split_kernel (+ wgrad_reduce_kernel)"}.get(dk, dk),
                   "launches_per_step": round(len(fam[dom_key]) / n_rec, 2), "launch_ms_mean": round(dmean, 5),
                   "algorithmic_gflop_per_launch": round(dfl / 1e9, 3), "algorithmic_mb_per_launch": round(dby / 1e6, 2),
                   "issued_tflops": round(issued_mult * dfl / (dmean * 1e-3) / 1e12, 1),
                   "frac": round(issued_mult * dfl / (dmean * 1e-3) / 1e12 / issued_peak, 4), "traffic": None, "traffic_ratio": None}
            tr = pmc_k.get(f"{dk}:{int(dfl)}:{int(dby)}")
            if tr:
                dom["traffic"] = tr
                dom["traffic_ratio"] = round(tr / dby, 3) if dby else None
        # the family's HBM traffic: PMC bytes (2 FETCH_SIZE + WRITE_SIZE) of every member launch of a step, matched by launch shape
        fam_traffic = fam_traffic_ratio = fam_traffic_tw = None
        fam_cov = 0
        if pmc_k and fam:
            tb = ab = tw_num = tw_den = 0.0
            for (kind, fl, by), v in fam.items():
                tr = pmc_k.get(f"{kind}:{int(fl)}:{int(by)}")
                if tr and by:
                    fam_cov += len(v)
                    tb += tr * len(v)
                    ab += by * len(v)
                    tw_num += (tr / by) * sum(v)
                    tw_den += sum(v)
            if fam_cov:
                fam_traffic = tb / fam_cov                   # mean PMC bytes per launch of the family
                fam_traffic_ratio = round(tb / ab, 3)        # all PMC bytes / all algorithmic bytes of the matched launches
                fam_traffic_tw = round(tw_num / tw_den, 3)   # per-launch ratios weighted by launch time
        roofline_family = {
            "kernel": ("MLP matrix-kernel family: gemm_f16_kernel / gemm_split_kernel (activation GEMMs with ReLU / mask / reconstruction-"
                       "loss epilogues) + wgrad_split_jobs_kernel / wgrad_split_kernel (weight gradients, several layers per launch), " f"{round(fam_launches / max(n_rec, 1), 1)} launches per step"
                       if args.mlp != "library" else "library fp32 GEMMs are not recorded; fp32-MFMA weight gradients only"),
            "bound": "mfma",
            "achieved": round(issued_mult * fam_alg, 2), "peak": issued_peak, "unit": "TFLOP/s",
            "frac": round(issued_mult * fam_alg / issued_peak, 4),
            "frac_kind": f"ISSUED matrix-instruction FLOPs ({issued_mult:g} fp16 piece products per fp32 product) / time of the family's launches "
                         f"in the first {n_rec} steps of the timed region (HIP events on the launch stream) / dense fp16 MFMA peak",
            "algorithmic_tflops": round(fam_alg, 2), "algorithmic_frac_of_issued_peak": round(fam_alg / issued_peak, 4),
            "algorithmic_frac_of_fp32_peak": round(fam_alg / PEAK_FP32_MFMA_TFLOPS, 4),
            "family_ms_per_step": round(fam_ms / max(n_rec, 1), 4), "share_of_step": round(fam_ms / max(n_rec, 1) / ms_per_step, 3),
            "recorded_steps": n_rec,
            "launches": fam_launches,
            "traffic": fam_traffic,
            "traffic_ratio": fam_traffic_ratio, "traffic_ratio_time_weighted": fam_traffic_tw,
            "traffic_source": (f"{pmc_k_src}: mean HBM bytes per launch over the family's {fam_launches} launches ({fam_cov} matched by shape), "
                               "2*FETCH_SIZE+WRITE_SIZE of separate rocprofv3 --pmc passes on the same librqhip.so; traffic_ratio = PMC bytes / "
                               "algorithmic bytes over those launches, _time_weighted = per-launch ratios weighted by launch time"
                               if fam_cov else pmc_k_src),
            "dominant_member": dom,
        }
        workload = {
            "c2": "C2: synthetic 100000x768 unit-norm items per GPU -> RQ-VAE 768-[512,256,128]-32, 3x256 codebooks, "
                  "STE (Gumbel off), one fwd+bwd+allreduce+AdamW step per HBM-resident batch",
            "c4": "C4: synthetic 10Mx768 items sharded over 8 GPUs = 1250000 rows per GPU (seed 1234+rank) -> RQ-VAE "
                  "768-[512,256,128]-32, 4x1024 codebooks, STE; one step = the whole shard as 10 micro-batches of "
                  "125000 rows (gradient accumulation), one flat gradient all-reduce, one AdamW update",
        }[args.config]
        line = {
            "metric": "item-embeddings quantized/sec (RQ-VAE fwd+bwd)",
            "value": round(value, 1), "unit": "items/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"split": "f32 (MLP GEMMs f16x2-emulated)", "split6": "f32 (MLP GEMMs bf16x3-emulated)", "library": "f32"}[args.mlp],
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "rows_per_gpu_per_step": B, "micro_batch_rows": micro,
                       "levels": LEVELS, "codebook_size": CODES, "embed_dim": EMBED,
                       "parallelism": f"row-shard x{world}, 1 flat grad all-reduce per step (RCCL)"},
            "roofline": roofline_family,
            "roofline_rq": {"kernel": (f"rq_seam_kernel<STE> = 128->{EMBED} GEMM + rq_forward's filtered scan ({LEVELS}x{CODES}) + {EMBED}->128 GEMM + ReLU, "
                                       f"{Bm} rows/launch (csrc/rq_forward.hip)" if seam_fused
                                       else f"rq_forward_kernel<16,STE,filtered> ({LEVELS}x{CODES}, {Bm} rows/launch)"), "bound": "mfma",
                         "measured": f"HIP events over {n_prof} steps run after the timed region (the timed region's records are the GEMM family's)",
                         "binding_resource": "valu-epilogue: the (best, runner-up, index) tournament on the 16 scores a lane "
                                             "receives per 32 codes -- neither matrix pipe nor HBM limits the kernel; "
                                             "`bound` names the FLOP roofline the fraction is priced on (the contract's "
                                             "two-value field)",
                         "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "frac_kind": "algorithmic fp32 FLOPs (L(2DK+5D) per row) / dense fp32 MFMA peak; the scan issues "
                                      "bf16 matrix instructions, so values above 1 are possible",
                         "traffic": traffic,
                         "traffic_source": traffic_src, "launch_ms_mean": round(mean_ms, 5),
                         "launches": len(kernel_ms), "flops_per_row": rq_flops_row, "fused_seam": seam_fused,
                         "scan": {"arithmetic": "scores x.c - |c|^2/2 from a 3-term bf16 split of the fp32 operands plus the "
                                                "exact bf16 pieces of -|c|^2/2 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); "
                                                "every row whose two best scores are within the proven error bound "
                                                "(tests/test_filter_bound.py) is re-decided with the oracle's fp32 FMA chain; "
                                                "ids, losses and outputs bit-identical to the oracle and to the all-fp32 kernel "
                                                "(parity.product_kernel_equals_margin_kernel)",
                                  "issued_bf16_tflops": round(LEVELS * (3 * 2 * EMBED + 2 * 16) * CODES * Bm / (mean_ms * 1e-3) / 1e12, 2),
                                  "frac_of_bf16_peak": round(LEVELS * (3 * 2 * EMBED + 2 * 16) * CODES * Bm / (mean_ms * 1e-3) / 1e12
                                                             / PEAK_BF16_MFMA_TFLOPS, 4),
                                  "all_fp32_kernel": {"flag": "RQHIP_FWD_SCAN_FP32 (rqhip_rq_forward_ex)",
                                                      "launch_ms_mean": round(f32_mean, 5), "launches": len(f32_ms),
                                                      "frac": round(flops_per_row * Bm / (f32_mean * 1e-3) / 1e12
                                                                    / PEAK_FP32_MFMA_TFLOPS, 4)}},
                         "hbm_view": {"algorithmic_bytes_per_row": bytes_per_row,
                                      "achieved_GBps": round(bytes_per_row * Bm / (mean_ms * 1e-3) / 1e9, 1),
                                      "peak_GBps": PEAK_HBM_GBPS},
                         "whole_step": {"flops_per_row": step_flops_per_row, "achieved_TFLOPs": round(step_tflops / world, 2),
                                        "issued_dtype_frac": roofline_kernels["matrix_kernels"]["frac_of_issued_dtype_peak"],
                                        "fp32_equivalent_frac": round(step_tflops / PEAK_FP32_MFMA_TFLOPS / world, 4),
                                        "note": "issued_dtype_frac: matrix-instruction FLOPs of the GEMM kernels / their time / "
                                                "the dense peak of the dtype they issue (see roofline_kernels); "
                                                "fp32_equivalent_frac: all GEMM + RQ FLOPs of fwd+bwd per GPU / ms_per_step / "
                                                "157.3 -- above 1 is possible, the kernels do not use the fp32 pipe"}},
            "roofline_kernels": roofline_kernels,
            "arithmetic": {"split": "f16x2: MLP GEMMs and weight gradients as two fp16 pieces per fp32 operand (11 + 11 significant "
                                    "bits + sign) under exact power-of-two row / column scales, products hh + hm + mh on "
                                    "v_mfma_f32_32x32x16_f16, fp32 accumulation -- an emulation of fp32 GEMMs held to `max error "
                                    "vs fp64 <= the library fp32 GEMM's` on every operand family of tests/test_gpu_gemm_split.py; "
                                    "the RQ kernels are fp32 (bit-exact vs the oracle)",
                           "split6": "bf16x3: three exact bf16 pieces per operand, six products (round 3)",
                           "library": "fp32 library GEMMs + fp32-MFMA weight gradients (round 2)"}[args.mlp],
            "breakdown_ms": {"rows": Bm, "encoder_fwd": round(enc_ms, 3), "rq_forward_call": round(rq_ms, 3),
                             "model_fwd_total": round(fwd_ms, 3), "backward_total": round(bwd_ms, 3),
                             "allreduce_ms": round(allreduce_ms, 4), "adamw": round(opt_ms, 3),
                             "first_forward_with_kmeans_init_s": round(kmeans_s, 3),
                             "kmeans_init_s": round(kmeans_only_s, 4)},
            "rccl_ranks": 0 if dry else world,
            "dry_ranks": world if dry else 0,
            "allreduce": {"exposed_ms_per_rank_device_host": exposed_all,
                          "overlap_launches": reducer_overlaps,
                          "what": "per rank, mean over the timed steps: the part of the step's gradient reduction that is NOT hidden under the "
                                  "encoder's backward (wait for the early all-reduces + the late ones), by events on the compute stream and by the "
                                  "host clock; DESIGN.md section 6 predicts >= 0.96 weak-scaling efficiency at 8 GPUs from it",
                          "backend": ("gloo (dry run: ranks share cuda:0, tensors staged through the host)" if dry else "nccl (RCCL over xGMI)") if world > 1 else None},
            "secondary": {"strict_fp32": strict,
                          "s_rq_items_per_s": round(Bm / srq_ms * 1e3, 1), "s_rq_ms_fwd_bwd": round(srq_ms, 4),
                          "tokenize_items_per_s": round(Bm / tok_ms * 1e3, 1), "tokenize_ms": round(tok_ms, 4),
                          "note": "per GPU; S-rq = HIP quantisation stack fwd+bwd on 32-d latents, tokenize = "
                                  "get_semantic_ids (encoder GEMMs + HIP RQ, eval)"},
            "mlp_gemms": (f"--mlp {args.mlp} (see `arithmetic`); layers of 128 (mod 128) output columns on the split kernels"
                          + (" (128 (mod 256) on the library: --no-narrow)" if args.no_narrow else "")
                          + ("; the 32-wide layers either side of the quantiser: the seam kernel (rqhip_rq_seam: fused with the levels in the "
                             "forward, stand-alone with the ReLU backward / maxima epilogues as the data gradients; fp32 FMA chains), their weight "
                             "gradients fp32-MFMA kernels" if (not args.no_seam and micro >= 4096 and args.mlp != "library") else
                             "; the 32-wide layers: PyTorch-ROCm fp32 GEMMs (matmul precision highest), TunableOp selections "
                             + ("loaded" if tuned else "off"))),
            "input_scales": ("the per-row maxima the fp16-split GEMMs scale by come with the resident item matrix (computed once per corpus, "
                             "data/processed.py:ItemData._corpus_maxima; the first layer's weight gradient uses corpus-wide column bounds): no maxima "
                             "pass over the input batch inside the step" if (not args.no_input_scales and args.mlp == "split" and micro >= 4096)
                             else "a maxima pass over the input batch inside every step (round 4; --no-input-scales)"),
            "final_loss": round(final_loss, 6), "p_unique_ids": round(p_unique, 6),
            "librqhip_sha256": lib_sha,
        }
        if long_run is not None:
            line["long_run"] = long_run
        if dry:
            line["config"]["parallelism"] = f"DRY RUN: {world} ranks on ONE GPU over gloo -- plumbing rehearsal, not a scaling number"
            line["n_physical_gpus"] = 1
        del model, opt, reducer, batches, X
        torch.cuda.empty_cache()
        line["box"] = box_calibration(device)
        if not args.no_parity:
            line["parity"] = parity_gate(device, args.config)
            if args.config != "c4":     # the C4-shaped fixture (4 x 1024 codebooks, 300 000 rows end to end) beside it
                try:
                    # ... under BOTH arithmetics of the MLP GEMMs in this one process: the product's f16x2 emulation and
                    # strict fp32 (library GEMMs) -- does the emulation add id flips against the reference?
                    p4 = {}
                    for arm, arith in (("f16x2", "f16x2"), ("fp32", "fp32")):
                        prev = _lin.use_arith(arith)
                        try:
                            p4[arm] = parity_gate(device, "c4")
                        finally:
                            _lin.use_arith(prev)
                    pa = p4["f16x2"] if args.mlp != "library" else p4["fp32"]
                    line["parity"]["c4"] = {
                        "rows_total": pa["end_to_end"]["eval"]["rows_total"], "mismatches_eval": pa["end_to_end"]["eval"]["mismatches"],
                        "mismatches_train": pa["end_to_end"]["train"]["mismatches"], "ids_exact_rate": pa["ids_exact_rate"],
                        "mismatch_margins_eval": pa["end_to_end"]["eval"]["mismatch_margins"],
                        "hard_rows_mismatches": pa["kernel_level_hard_rows"]["mismatches"],
                        "kernel_level_mismatches": pa["kernel_level"]["mismatches"],
                        "all_mismatches_flagged": pa["all_mismatches_flagged"], "loss_max_abs_err": pa["loss_max_abs_err"],
                        "pass": pa["pass"],
                        "mismatches_f16x2": {"eval": p4["f16x2"]["end_to_end"]["eval"]["mismatches"], "train": p4["f16x2"]["end_to_end"]["train"]["mismatches"],
                                             "all_flagged": p4["f16x2"]["all_mismatches_flagged"], "pass": p4["f16x2"]["pass"]},
                        "mismatches_fp32": {"eval": p4["fp32"]["end_to_end"]["eval"]["mismatches"], "train": p4["fp32"]["end_to_end"]["train"]["mismatches"],
                                            "all_flagged": p4["fp32"]["all_mismatches_flagged"], "pass": p4["fp32"]["pass"]},
                        "arms": "the same fixture end to end with the MLP GEMMs as f16x2 split kernels (the product) and as library fp32 GEMMs "
                                "(bench.py --mlp library), same process; tau (tests/parity_gate.py) 2e-6 end to end"}
                    line["parity"]["c4_mismatches"] = pa["mismatches"]
                except Exception as e:  # noqa: BLE001  (a missing fixture must not cost the line)
                    line["parity"]["c4"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_rows, LEVELS, CODES)
        if world == 1 and not args.no_small_batch:
            line["secondary"]["small_batch"] = small_batch_secondary(line.get("cpu_baseline"))
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    rqdist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
