"""CPU-only tests of the host side: gin subset, configs, module surface, data stand-in, loud failure
without a GPU.  No kernel is launched here."""
import os

import pytest
import torch

from conftest import PKG


def test_ginlite_parses_the_shipped_configs():
    from rqhip import ginlite
    import data.processed  # noqa: F401  (registers %data.processed.RecDataset.*)
    import modules.quantize as mq
    ginlite.clear_config()
    ginlite.parse_config_file(os.path.join(PKG, "configs", "rqvae_amazon.gin"))
    assert ginlite.query_parameter("train.batch_size") == 640
    assert ginlite.query_parameter("train.vae_hidden_dims") == [512, 256, 128]
    assert ginlite.query_parameter("train.learning_rate") == 0.001
    assert ginlite.query_parameter("train.vae_codebook_mode") is mq.QuantizeForwardMode.STE
    assert ginlite.query_parameter("train.dataset").name == "AMAZON"
    assert ginlite.query_parameter("train.save_dir_root") == "out/rqvae/amazon/"
    ginlite.clear_config()
    ginlite.parse_config_file(os.path.join(PKG, "configs", "rqvae_ml32m.gin"))
    assert ginlite.query_parameter("train.vae_embed_dim") == 64
    assert ginlite.query_parameter("train.vae_codebook_mode") is mq.QuantizeForwardMode.ROTATION_TRICK
    ginlite.clear_config()


def test_ginlite_configurable_binding_precedence_and_errors():
    from rqhip import ginlite

    @ginlite.configurable
    def fn(a=1, b="x", c=None):
        return a, b, c

    ginlite.clear_config()
    ginlite.parse_config("# comment\nfn.a = 5\nfn.b = 'has # hash'  # trailing\n")
    assert fn() == (5, "has # hash", None)
    assert fn(a=7) == (7, "has # hash", None)
    assert fn(9) == (9, "has # hash", None)
    ginlite.parse_config("fn.nope = 1")
    with pytest.raises(ValueError):
        fn()
    ginlite.clear_config()
    with pytest.raises(ValueError):
        ginlite.parse_config("fn.a = %not.a.Constant")
    with pytest.raises(ValueError):
        ginlite.parse_config("just words")
    ginlite.clear_config()


def test_train_signature_matches_reference_kwargs():
    import inspect
    import train_rqvae
    fn = inspect.unwrap(train_rqvae.train)
    names = list(inspect.signature(fn).parameters)
    expected = ["iterations", "batch_size", "learning_rate", "weight_decay", "dataset_folder", "dataset",
                "pretrained_rqvae_path", "save_dir_root", "use_kmeans_init", "split_batches", "amp", "wandb_logging",
                "do_eval", "force_dataset_process", "mixed_precision_type", "gradient_accumulate_every",
                "save_model_every", "eval_every", "commitment_weight", "vae_n_cat_feats", "vae_input_dim",
                "vae_embed_dim", "vae_hidden_dims", "vae_codebook_size", "vae_codebook_normalize", "vae_codebook_mode",
                "vae_sim_vq", "vae_n_layers", "dataset_split"]
    assert names[:29] == expected   # reference train_rqvae.py:25-55, same order and names


def test_rqvae_surface_and_state_dict_keys():
    from modules.quantize import Quantize, QuantizeForwardMode, QuantizeOutput
    from modules.rqvae import RqVae, RqVaeComputedLosses, RqVaeOutput
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256,
              codebook_mode=QuantizeForwardMode.STE, n_layers=3, n_cat_features=0)
    keys = set(m.state_dict())
    expected = {f"layers.{l}.embedding.weight" for l in range(3)}
    expected |= {f"encoder.mlp.{i}.weight" for i in (0, 2, 4, 6)} | {f"decoder.mlp.{i}.weight" for i in (0, 2, 4, 6)}
    assert keys == expected
    assert sum(p.numel() for p in m.parameters()) == 1146880   # 1 146 880 fp32 grads
    assert m.layers[0].weight.shape == (256, 32)
    assert 0.0 <= float(m.layers[0].weight.min()) and float(m.layers[0].weight.max()) <= 1.0  # uniform(0,1) init
    assert m.config["embed_dim"] == 32 and m.config["codebook_size"] == 256
    assert RqVaeOutput._fields == ("embeddings", "residuals", "sem_ids", "quantize_loss")
    assert RqVaeComputedLosses._fields == ("loss", "reconstruction_loss", "rqvae_loss", "embs_norm", "p_unique_ids")
    assert QuantizeOutput._fields == ("embeddings", "ids", "loss")
    q = Quantize(embed_dim=8, n_embed=4)
    assert q.forward_mode is QuantizeForwardMode.GUMBEL_SOFTMAX and q.do_kmeans_init and not q.kmeans_initted
    assert q.get_item_embeddings(torch.tensor([1, 3])).shape == (2, 8)


def test_seeded_construction_order_matches_reference_fixture():
    """Parameters are created in the reference's order, so torch.manual_seed gives the same init:
    checked against the weights stored in a golden fixture (seed 31, rqvae_small_ste)."""
    from conftest import load_golden
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    g = load_golden("rqvae_small_ste.npz")
    torch.manual_seed(31)
    m = RqVae(input_dim=48, embed_dim=16, hidden_dims=[32, 24], codebook_size=32, n_layers=3, n_cat_features=0,
              codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE)
    for k, v in m.state_dict().items():
        if "embedding" in k:
            continue  # the fixture overwrote the codebooks after construction
        assert torch.equal(v, torch.from_numpy(g["param::" + k])), k


def test_cpu_tensors_fail_loudly_no_fallback():
    from modules.quantize import Quantize, QuantizeForwardMode
    from rqhip import RqHipError
    q = Quantize(embed_dim=8, n_embed=4, do_kmeans_init=False, forward_mode=QuantizeForwardMode.STE)
    with pytest.raises(RqHipError, match="no CPU fallback"):
        q(torch.randn(3, 8), temperature=0.2)
    from init.kmeans import kmeans_init_
    np_seed = __import__("numpy").random.seed
    np_seed(0)
    with pytest.raises(RqHipError):
        kmeans_init_(torch.zeros(4, 8), torch.randn(16, 8))


def test_decoder_side_consumers_have_no_cpu_path_and_keep_the_reference_signatures():
    """evaluate/metrics.py:7-28 and modules/model.py:169-182 mirrors: same constructor / method names and argument
    checks as the reference; the matching itself is HIP only."""
    from evaluate.metrics import TopKAccumulator
    from modules.sid_prefix import SemIdPrefixIndex
    from rqhip import RqHipError
    acc = TopKAccumulator()
    assert acc.ks == [1, 5, 10] and acc.total == 0 and acc.reduce.__name__ == "reduce"
    with pytest.raises(RqHipError, match="no CPU fallback"):
        acc.accumulate(actual=torch.zeros((2, 3), dtype=torch.int64), top_k=torch.zeros((2, 4, 3), dtype=torch.int64))
    with pytest.raises(ValueError):
        SemIdPrefixIndex(torch.zeros((4, 3), dtype=torch.int32))
    index = SemIdPrefixIndex(torch.arange(12).reshape(4, 3))         # CPU codebooks: index is built on first use
    assert index.num_items == 4 and index._index is None
    with pytest.raises(RqHipError, match="no CPU fallback"):
        index.check_valid_prefix(torch.tensor([[0, 1]]))


def test_mlp_fused_linear_relu_function_matches_plain_ops():
    """modules/encoder.py: the autograd wrapper around the ReLU-epilogue GEMM (GPU path) computes what Linear + ReLU
    compute; checked here on CPU, where aten::_addmm_activation also exists."""
    from modules.encoder import MLP, _LinearReLU
    torch.manual_seed(3)
    x = torch.randn(9, 8, dtype=torch.float64, requires_grad=True)
    w = torch.randn(5, 8, dtype=torch.float64, requires_grad=True)
    zero = torch.zeros(5, dtype=torch.float64)
    assert torch.autograd.gradcheck(lambda a, b: _LinearReLU.apply(a, b, zero), (x, w))
    y = _LinearReLU.apply(x, w, zero)
    assert torch.allclose(y, torch.relu(x @ w.t()))
    m = MLP(8, [6, 4], 3)                                              # CPU tensors take the plain nn.Sequential path
    out = m(torch.randn(5, 8))
    assert out.shape == (5, 3) and list(m.state_dict()) == ["mlp.0.weight", "mlp.2.weight", "mlp.4.weight"]


def test_quantize_asserts_and_unsupported_modes():
    from modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode
    q = Quantize(embed_dim=8, n_embed=4, do_kmeans_init=False, forward_mode=QuantizeForwardMode.STE)
    with pytest.raises(AssertionError):
        q(torch.randn(3, 7), temperature=0.2)
    from rqhip import RqHipError
    qc = Quantize(embed_dim=8, n_embed=4, do_kmeans_init=False, distance_mode=QuantizeDistance.COSINE,
                  forward_mode=QuantizeForwardMode.STE)
    with pytest.raises(RqHipError):          # implemented, but GPU only like everything else
        qc(torch.randn(3, 8), temperature=0.2)
    # shapes / combinations outside the kernels run as PyTorch-ROCm operators ON THE DEVICE (rqhip/wide.py): no CPU path either
    from rqhip import wide
    assert wide.stack_covers(128, 65536, 16) and not wide.stack_covers(129, 256) and not wide.stack_covers(32, 256, 17)
    assert wide.gumbel_covers(32, 256) and wide.gumbel_covers(64, 256) and not wide.gumbel_covers(8, 1100) and not wide.gumbel_covers(160, 40)
    for kw in (dict(embed_dim=160, n_embed=4, forward_mode=QuantizeForwardMode.STE),
               dict(embed_dim=8, n_embed=4, forward_mode=QuantizeForwardMode.GUMBEL_SOFTMAX, distance_mode=QuantizeDistance.COSINE)):
        qw = Quantize(do_kmeans_init=False, **kw)
        assert not qw.kernel_covers()
        with pytest.raises(RqHipError):
            qw(torch.randn(3, kw["embed_dim"]), temperature=0.2)


def test_item_data_contract():
    from data.processed import ItemData, RecDataset, synthetic_item_matrix, synthetic_train_mask
    X = torch.cat([synthetic_item_matrix(50), torch.ones(50, 10)], dim=1)   # wider than 768: sliced
    mask = synthetic_train_mask(50)
    tr = ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split="train", item_matrix=X, is_train=mask)
    ev = ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split="eval", item_matrix=X, is_train=mask)
    al = ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split="all", item_matrix=X, is_train=mask)
    assert len(tr) + len(ev) == len(al) == 50
    b = al[torch.tensor([3, 1, 4])]
    assert b.x.shape == (3, 768) and torch.equal(b.x, X[[3, 1, 4], :768])
    assert torch.equal(b.ids, torch.tensor([3, 1, 4])) and b.seq_mask.dtype == torch.bool
    b = al[[5, 6]]
    assert b.ids.shape == (1, 2) and b.x.shape == (2, 768)      # list index -> ids [1, n] (processed.py:75-77)
    b = al[7]
    assert b.ids.shape == (1,) and b.x.shape == (768,)
    norms = synthetic_item_matrix(20).norm(dim=1)
    assert torch.allclose(norms, torch.ones(20), atol=1e-5)


def test_device_batcher_epochs_cover_dataset():
    from data.processed import ItemData, synthetic_item_matrix
    import train_rqvae
    ds = ItemData(root="/nonexistent", item_matrix=synthetic_item_matrix(23), train_test_split="all")
    g = torch.Generator().manual_seed(0)
    it = train_rqvae._DeviceBatcher(ds, 10, generator=g)
    seen = torch.cat([next(it).ids for _ in range(3)])
    assert sorted(seen.tolist()) == list(range(23))          # 10 + 10 + 3, no replacement within an epoch
    assert next(it).ids.numel() == 10                        # next epoch starts


def test_quantize_loss_definition_and_gpu_only_recon_loss():
    from modules.loss import CategoricalReconstuctionLoss, QuantizeLoss, ReconstructionLoss
    from rqhip import RqHipError
    a, b = torch.randn(5, 12), torch.randn(5, 12)
    ql = QuantizeLoss(0.25)(a, b)
    assert torch.allclose(ql, 1.25 * ((a - b) ** 2).sum(-1), rtol=1e-6)
    with pytest.raises(RqHipError, match="no CPU fallback"):
        ReconstructionLoss()(a, b)
    with pytest.raises(RqHipError):
        CategoricalReconstuctionLoss(4)(a, b)


def test_entry_point_fails_loudly_without_gpu():
    """`python train_rqvae.py <cfg.gin>` (BASELINE config 1, plumbing): config parsing, enum constants and the
    train() binding work on a CPU-only box; the run itself stops with a clear error, not a silent CPU fallback."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_train.py")
    proc = subprocess.run([sys.executable, os.path.join(PKG, "train_rqvae.py"), os.path.join(PKG, "configs", "rqvae_amazon.gin")],
                          cwd=PKG, capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    assert "needs a ROCm GPU" in proc.stderr


def test_item_data_refuses_to_invent_a_corpus(tmp_path, monkeypatch, capsys):
    """a missing item_features.pt raises; synthetic items need the explicit "synthetic:<n>" folder name."""
    import pytest
    from data.processed import ItemData
    with pytest.raises(FileNotFoundError, match="item_features.pt"):
        ItemData(root=str(tmp_path / "dataset" / "amazon"))
    ds = ItemData(root="synthetic:40")
    assert ds.synthetic and len(ds) == 40
    assert "SYNTHETIC" in capsys.readouterr().out


def test_bench_relaunches_itself_under_torchrun_for_multi_gpu(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun environment must start its own N ranks."""
    import importlib
    import os
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--config", "c4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit:
        pass
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-6:] == ["--gpus", "4", "--config", "c4", "--steps", "2"] and a[-7].endswith("bench.py")
    # and inside a torchrun environment it does not relaunch: the world size must simply match --gpus
    assert bench.CONFIGS["c4"]["rows"] == 1_250_000 and bench.CONFIGS["c4"]["codes"] == 1024


def test_loss_scale_hint_nests_and_restores():
    """rqhip.autograd.loss_scale: the hint for the speculative reconstruction-loss gradient is a plain nesting context
    manager -- restored on exit and on exceptions (a stale value would only cost speed, but it must not leak)."""
    from rqhip import autograd as ag
    assert ag._LOSS_SCALE == 1.0
    with ag.loss_scale(0.25):
        assert ag._LOSS_SCALE == 0.25
        with ag.loss_scale(0.5):
            assert ag._LOSS_SCALE == 0.5
        assert ag._LOSS_SCALE == 0.25
    assert ag._LOSS_SCALE == 1.0
    try:
        with ag.loss_scale(0.1):
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert ag._LOSS_SCALE == 1.0


def test_grad_sink_is_claimed_once_per_parameter_and_epoch():
    """rqhip.dist.claim_grad_sink: the in-place slice goes to the first producer of an epoch only."""
    import torch
    from rqhip.dist import FlatGradReducer, claim_grad_sink
    w = torch.nn.Parameter(torch.zeros(3, 2))
    other = torch.nn.Parameter(torch.zeros(2))
    assert claim_grad_sink(w) is None                      # never attached
    red = FlatGradReducer([w, other]).attach(None)
    red.zero_()
    v = claim_grad_sink(w)
    assert v is not None and v.data_ptr() == red.flat.data_ptr()
    assert claim_grad_sink(w) is None                      # second producer of the same backward pass
    assert claim_grad_sink(other) is not None              # claims are per parameter
    red.zero_()
    assert claim_grad_sink(w) is not None                  # released by the next epoch
    red.zero_()
    w.grad = torch.ones(3, 2)
    assert claim_grad_sink(w) is None                      # a gradient is already there: autograd accumulates


def test_small_batch_weight_gradient_routing_and_its_abi_argument_checks():
    """Which batches take the job-table weight-gradient kernel (csrc/wgrad_jobs.hip) is host logic: below the split kernels' 4096
    rows, at most eight supported layers, not under the strict fp32 arithmetic; and the C entry point refuses bad job tables before
    it touches a device (no GPU here)."""
    import ctypes as C
    from rqhip import _lib, linear, ops
    mlp = [(512, 768), (256, 512), (128, 256), (32, 128)]
    assert linear.wgrad_jobs_ok(640, mlp) and linear.wgrad_jobs_ok(64, mlp[::-1]) and linear.wgrad_jobs_ok(4095, mlp)
    assert not linear.wgrad_jobs_ok(4096, mlp) and not linear.wgrad_jobs_ok(100_000, mlp) and not linear.wgrad_jobs_ok(0, mlp)
    assert not linear.wgrad_jobs_ok(640, mlp + [(48, 64)])            # a layer no block shape tiles
    assert not linear.wgrad_jobs_ok(640, mlp * 3) and not linear.wgrad_jobs_ok(640, [])
    before = linear.use_arith("fp32")
    try:
        assert not linear.wgrad_jobs_ok(640, mlp)                     # the oracle-ordered kernels keep that arithmetic to themselves
    finally:
        linear.use_arith(before)
    off = linear.use_wgrad_jobs(False)
    try:
        assert not linear.wgrad_jobs_ok(640, mlp)
    finally:
        linear.use_wgrad_jobs(off)
    assert ops.linear_wgrad_jobs_supported(32, 32) and not ops.linear_wgrad_jobs_supported(16, 64)
    l = _lib.lib()
    vp, ci = C.c_void_p * 9, C.c_int * 9
    fake = vp(*[0x1000] * 9)                                          # never dereferenced: every call below is refused first
    assert l.rqhip_linear_wgrad_jobs(fake, fake, fake, ci(*[64] * 9), ci(*[64] * 9), 9, 640, None) == -1   # RQHIP_EARG
    assert l.rqhip_linear_wgrad_jobs(fake, fake, fake, ci(*[64] * 9), ci(*[64] * 9), -1, 640, None) == -1   # RQHIP_EARG
    assert l.rqhip_linear_wgrad_jobs(fake, fake, fake, ci(*[64] * 9), ci(*[64] * 9), 2, -5, None) == -1   # RQHIP_EARG
    assert l.rqhip_linear_wgrad_jobs(fake, fake, fake, ci(*([48] + [64] * 8)), ci(*[64] * 9), 2, 640, None) == -1   # RQHIP_EARG
    assert l.rqhip_linear_wgrad_jobs(vp(*([0] + [0x1000] * 8)), fake, fake, ci(*[64] * 9), ci(*[64] * 9), 2, 640, None) == -1   # RQHIP_EARG
    assert l.rqhip_linear_wgrad_jobs(None, None, None, None, None, 0, 640, None) == 0      # no jobs: nothing to do
    with pytest.raises(_lib.RqHipError):
        ops.linear_wgrad_jobs([(torch.zeros(8, 64), torch.zeros(8, 64))])                              # CPU tensors: no fallback


def test_flat_grad_reducer_pads_every_slice_to_16_bytes():
    """a parameter whose numel is not a multiple of 4 must not misalign the slices behind it (csrc/adamw.hip and the
    weight-gradient sinks read 16 bytes at a time)."""
    from rqhip.dist import FlatGradReducer
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((3,), (5, 1), (2, 4), (7,), (4, 4))]
    red = FlatGradReducer(ps)
    assert red._offsets == [0, 4, 12, 20, 28] and red.flat.numel() == 44
    for p, v in zip(ps, red._views):
        assert v.shape == p.shape and v.storage_offset() % 4 == 0
    # contiguous runs cover the padding (one all-reduce per run); a run never splits a parameter
    assert red._runs([0, 1, 2]) == [(0, 20)] and red._runs([0, 2]) == [(0, 4), (12, 20)] and red._runs([3, 4]) == [(20, 44)]
    for v in red._views:
        v.fill_(1.0)
    assert float(red.flat.sum()) == 3 + 5 + 8 + 7 + 16          # the padding floats stay zero
