"""SURVEY.md section 8 row f3 on the GPU: the item-feature matrix resident in HBM, batches gathered on the device
(reference data/processed.py:39-86 keeps it in host RAM and indexes `[idx, :768]` per batch, data/utils.py:10-11 copies)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _datasets(n, extra=10, seed=5):
    from data.processed import ItemData, RecDataset, synthetic_item_matrix, synthetic_train_mask
    X = torch.cat([synthetic_item_matrix(n, seed=seed), torch.randn(n, extra)], dim=1)      # wider than 768: sliced
    mask = synthetic_train_mask(n)
    mk = lambda split: ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split=split, item_matrix=X, is_train=mask)
    return X, mask, mk


def test_device_gather_equals_host_indexing_bits():
    """`ds.to_device(dev)[idx]` == the reference's host-side `x[idx, :768]` bit for bit, for tensor, list and unsorted /
    repeated indices, on every split; ids, masks and the -1 placeholders as data/processed.py:72-86."""
    X, mask, mk = _datasets(5000)
    g = torch.Generator().manual_seed(1)
    for split, keep in (("all", torch.ones_like(mask)), ("train", mask), ("eval", ~mask)):
        host, dev = mk(split), mk(split).to_device("cuda")
        assert len(host) == len(dev) == int(keep.sum()) and dev.item_data.is_cuda and dev.item_data.shape[1] == X.shape[1]
        ref_rows = X[keep]
        for idx in (torch.randint(0, len(host), (640,), generator=g), torch.tensor([3, 3, 0, len(host) - 1]), torch.arange(len(host))):
            b = dev[idx]
            assert b.x.is_cuda and b.x.dtype == torch.float32 and b.x.shape == (len(idx), 768)
            assert torch.equal(b.x.cpu().view(torch.int32), ref_rows[idx, :768].contiguous().view(torch.int32))
            assert torch.equal(b.x.cpu(), host[idx].x) and torch.equal(b.ids.cpu(), idx)
            assert b.seq_mask.dtype == torch.bool and bool(b.seq_mask.all()) and int(b.user_ids.max()) == -1
        b = dev[[5, 6]]
        assert b.ids.shape == (1, 2) and torch.equal(b.x.cpu(), ref_rows[[5, 6], :768])


def test_device_batcher_epochs_cover_every_item_once():
    """train_rqvae._DeviceBatcher == BatchSampler(RandomSampler(ds), bs, drop_last=False) under `cycle` (train_rqvae.py:82-89):
    every epoch is a permutation of the items, the short final batch is kept, batches are gathered on the device."""
    from train_rqvae import _DeviceBatcher
    _, _, mk = _datasets(1000)
    ds = mk("train").to_device("cuda")
    n, bs = len(ds), 64
    it = _DeviceBatcher(ds, bs, generator=torch.Generator().manual_seed(3))
    per_epoch = (n + bs - 1) // bs
    for _ in range(2):
        ids = []
        for j in range(per_epoch):
            b = next(it)
            assert b.x.is_cuda and b.x.shape[0] == (bs if j + 1 < per_epoch or n % bs == 0 else n % bs)
            assert torch.equal(b.x, ds.item_data[b.ids, :768])
            ids.append(b.ids.cpu())
        assert torch.equal(torch.sort(torch.cat(ids)).values, torch.arange(n))
    seen = torch.cat([b.ids.cpu() for b in it.epoch()])
    assert torch.equal(torch.sort(seen).values, torch.arange(n))


def test_ten_million_item_matrix_stays_resident_and_tokenises():
    """BASELINE config 4's corpus (10 M x 768 fp32 = 30.7 GB) as ONE device allocation (288 GB of HBM per GPU): row offsets
    beyond 2^32 elements gather correctly and feed the kernels."""
    free, _total = torch.cuda.mem_get_info()
    n, d = 10_000_000, 768
    if free < n * d * 4 + (8 << 30):
        pytest.skip(f"needs {n * d * 4 / 2**30:.0f} GiB of free HBM, have {free / 2**30:.0f}")
    from data.processed import ItemData, RecDataset
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    X = torch.empty((n, d), device="cuda")
    probe = torch.tensor([0, 1, 2_796_203, 5_592_406, n - 2, n - 1])          # element offsets up to 7.68e9 > 2^32
    vals = torch.nn.functional.normalize(torch.randn(len(probe), d, generator=torch.Generator().manual_seed(2)), dim=-1)
    X[probe.cuda()] = vals.cuda()
    ds = ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split="all", item_matrix=X,
                  is_train=torch.ones(n, dtype=torch.bool, device="cuda"))
    assert ds.item_data.data_ptr() == X.data_ptr() and len(ds) == n          # the 'all' split is the matrix itself: no second copy
    b = ds[probe]
    assert torch.equal(b.x.cpu(), vals)
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=4, n_cat_features=0,
              codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda().eval()
    with torch.no_grad():
        a = m.get_semantic_ids(b.x).sem_ids
        c = m.get_semantic_ids(vals.cuda()).sem_ids
    assert np.array_equal(a.cpu().numpy(), c.cpu().numpy())
    del X, ds, b
    torch.cuda.empty_cache()


def test_corpus_maxima_travel_with_big_batches_and_replace_the_input_pass():
    """data/processed.py: batches of >= 4096 rows carry their rows' maxima (gathered from the once-per-corpus pass) and corpus-wide column
    bounds (rqhip/linear.py:attach_scales); the encoder then runs no maxima pass over its input: same row exponents -> the forward has the
    bits of the plain path, the first layer's weight gradient (column bounds instead of the batch's column maxima) stays inside the gate."""
    from data.processed import ItemData, synthetic_item_matrix
    from modules.encoder import MLP
    from rqhip import linear as lin
    from rqhip import ops
    ds = ItemData("unused", item_matrix=torch.cat([synthetic_item_matrix(20_000) * torch.logspace(-3, 3, 20_000)[:, None],
                                                   torch.zeros(20_000, 3)], dim=1)).to_device("cuda")     # 771 columns: features = the first 768
    idx = torch.randperm(20_000)[:6000]
    b = ds[idx]
    sc = lin.attached_scales(b.x)
    assert sc is not None and tuple(sc.rows.shape) == (1, 6000) and tuple(sc.cols.shape) == (768,)
    assert torch.equal(sc.rows[0].view(torch.float32), b.x.abs().amax(dim=1))
    assert (sc.cols.view(torch.float32) >= b.x.abs().amax(dim=0)).all()
    assert lin.attached_scales(ds[idx[:100]].x) is None                     # small batches do not take the fp16 path: nothing attached
    torch.manual_seed(2)
    mlp = MLP(768, [512, 256, 128], 32).cuda()
    gout = torch.randn(6000, 32, device="cuda")
    calls = []
    real = ops.maxima

    def counting(a, *args, **kw):
        calls.append(tuple(a.shape))
        return real(a, *args, **kw)

    res = []
    for x in (b.x, b.x.clone()):                                           # with the attached maxima / without (a clone carries none)
        for p in mlp.parameters():
            p.grad = None
        calls.clear()
        ops.maxima = counting
        try:
            y = mlp(x)
            y.backward(gout)
        finally:
            ops.maxima = real
        res.append((y.detach().clone(), [p.grad.clone() for p in mlp.parameters()], list(calls)))
    (y1, g1, c1), (y2, g2, c2) = res
    assert (6000, 768) not in c1 and (6000, 768) in c2                     # the pass over the input batch is gone
    assert torch.equal(y1, y2)                                             # same row exponents, same bits
    for a, bb in zip(g1[1:], g2[1:]):
        assert torch.equal(a, bb)
    ref = g2[0].double()
    assert (g1[0].double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()     # first layer: column bounds instead of maxima
