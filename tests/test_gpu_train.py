"""End-to-end: the gin-driven entry point on the GPU (BASELINE configs 1 and 3 in miniature) -- k-means warm-up,
optimisation steps, eval, id-diversity statistics, checkpoint, resume."""
import os

import pytest
import torch

from conftest import PKG

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]   # (whole training loops)


def _run(cfg_name, tmp_path, monkeypatch, **overrides):
    import train_rqvae
    from rqhip import ginlite
    try:
        import gin  # noqa: F401
        pytest.skip("real gin-config present: bindings below are written for the in-tree subset")
    except ImportError:
        pass
    ginlite.clear_config()
    ginlite.parse_config_file(os.path.join(PKG, "configs", cfg_name))
    out_dir = str(tmp_path) + "/"
    kw = dict(iterations=12, eval_every=6, save_model_every=12, save_dir_root=out_dir, wandb_logging=False,
              dataset_folder="synthetic:3000", log_every=4)
    kw.update(overrides)
    res = train_rqvae.train(**kw)
    ginlite.clear_config()
    return res, out_dir


def test_train_amazon_config_short_run_checkpoint_and_resume(tmp_path, monkeypatch):
    import numpy as np
    torch.manual_seed(0)
    np.random.seed(0)
    res, out_dir = _run("rqvae_amazon.gin", tmp_path, monkeypatch)
    assert res["loss"] == res["loss"] and res["loss"] < 5.0          # finite, and far below the untrained ~1e1
    ckpt = os.path.join(out_dir, "checkpoint_11.pt")
    assert os.path.exists(ckpt)
    state = torch.load(ckpt, map_location="cpu", weights_only=False)
    # the reference's four keys; a run on the opt-in synthetic corpus (dataset_folder="synthetic:<n>", as here) adds the marker "data"
    assert set(state) == {"iter", "model", "model_config", "optimizer", "data"} and state["iter"] == 11
    assert state["data"] == "synthetic"
    assert {"layers.0.embedding.weight", "encoder.mlp.0.weight", "decoder.mlp.6.weight"} <= set(state["model"])
    assert state["model"]["layers.0.embedding.weight"].shape == (256, 32)
    # resume: start_iter = iter + 1, optimizer state restored, k-means init skipped
    res2, _ = _run("rqvae_amazon.gin", tmp_path, monkeypatch, pretrained_rqvae_path=ckpt, iterations=4,
                   save_model_every=1000, eval_every=1000)
    assert res2["loss"] == res2["loss"]


@pytest.mark.parametrize("accumulate", [1, 2])
def test_train_loop_at_a_split_kernel_batch(tmp_path, monkeypatch, accumulate):
    """The gin-driven loop at a batch the split kernels take (8192 rows): the seam launch, the static-schedule GEMMs, and the weight gradients of
    both MLP stacks in the batched launches (the decoder's waiting for the encoder's when the gradients live in the flat buffer; with gradient
    accumulation the later micro-batches take the per-stack launches).  The loss falls like the per-layer path's, step for step to rounding."""
    import numpy as np
    from rqhip import linear
    losses = {}
    for arm in ("batched", "per_layer"):
        torch.manual_seed(0)
        np.random.seed(0)
        before = linear.use_wgrad_batch(arm == "batched")
        try:
            res, _ = _run("rqvae_amazon.gin", tmp_path, monkeypatch, batch_size=8192, dataset_folder="synthetic:20000", iterations=8,
                          eval_every=1000, save_model_every=1000, gradient_accumulate_every=accumulate, use_hip_graph=False)
        finally:
            linear.use_wgrad_batch(before)
        assert not linear._XSTACK and not linear._XSMALL
        losses[arm] = res["loss"]
        assert res["loss"] == res["loss"] and res["loss"] < 5.0
    assert abs(losses["batched"] - losses["per_layer"]) <= 2e-4 * abs(losses["per_layer"])


def test_train_ml32m_hyperparameters_rotation_trick(tmp_path, monkeypatch):
    """configs/rqvae_ml32m.gin: D = 64, ROTATION_TRICK, batch 64, k-means init (train() default)."""
    import numpy as np
    torch.manual_seed(1)
    np.random.seed(1)
    res, out_dir = _run("rqvae_ml32m.gin", tmp_path, monkeypatch, iterations=10, eval_every=5, save_model_every=10)
    assert res["loss"] == res["loss"]
    assert os.path.exists(os.path.join(out_dir, "checkpoint_9.pt"))


def test_hip_graph_step_matches_eager(tmp_path, monkeypatch):
    """use_hip_graph=True (also the default below 4096 rows: `None` = auto) replays the captured steps (full batch and epoch-tail batch, one
    graph each) across eval / tokenisation / checkpoint excursions; the loss stays on the eager trajectory (replayed kernels run in the order of the
    capture's streams, loss means through the single-launch form: close, not bit-equal)."""
    import numpy as np
    runs = []
    for flag in (False, True):
        torch.manual_seed(7)
        np.random.seed(7)
        res, _ = _run("rqvae_amazon.gin", tmp_path, monkeypatch, iterations=40, eval_every=16, save_model_every=1000,
                      do_eval=True, log_every=1, use_hip_graph=flag, batch_size=500)
        runs.append(res)
    assert abs(runs[0]["loss"] - runs[1]["loss"]) < 2e-3 * max(1.0, abs(runs[0]["loss"])), runs


def test_hip_graph_survives_hundreds_of_replays_interleaved_with_eager_work(tmp_path, monkeypatch):
    """Round-2 defect: the graph mode faulted after ~250 iterations.  Cause: hipMemsetAsync calls of the library (the id
    statistics) became memset NODES of the captured graph, and those broke replay once eager launches were interleaved
    (tools/graph_piece_probe.py); every memset is now a kernel (csrc/capi.hip:fill_words).  600 graphed iterations of
    the config-3 loop, eager gathers and copies between all of them."""
    import numpy as np
    torch.manual_seed(3)
    np.random.seed(3)
    res, _ = _run("rqvae_ml32m.gin", tmp_path, monkeypatch, iterations=600, eval_every=600, save_model_every=10 ** 6,
                  log_every=200, use_hip_graph=True)
    assert res["loss"] == res["loss"] and res["loss"] < 5.0


def test_both_step_shapes_of_an_epoch_are_replayed_and_captured_once(tmp_path, monkeypatch):
    """3000 synthetic items, 95 % of them in the training split = 2850 rows at batch 500: five full batches and one of 350 rows per epoch.
    Round 6 captures BOTH shapes (all warm-up steps before the first capture) and replays them: 60 iterations = 10 epochs with no eval /
    checkpoint excursion in between must cost exactly one capture per shape (round 5: one re-capture per epoch), and stay on the eager
    trajectory."""
    import numpy as np
    runs = []
    for flag in (False, True):
        torch.manual_seed(11)
        np.random.seed(11)
        res, _ = _run("rqvae_amazon.gin", tmp_path, monkeypatch, iterations=60, eval_every=10 ** 6, save_model_every=10 ** 6,
                      do_eval=True, log_every=1, use_hip_graph=flag, batch_size=500)
        runs.append(res)
    assert runs[0]["graph_captures"] == {}
    caps = runs[1]["graph_captures"]
    # (two each: the first capture, and one behind the eval / tokenisation pass that precedes the run's final step)
    assert len(caps) == 2 and 500 in caps and all(v == 2 for v in caps.values()), caps
    assert abs(runs[0]["loss"] - runs[1]["loss"]) < 2e-3 * max(1.0, abs(runs[0]["loss"])), runs
