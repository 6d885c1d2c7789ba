"""rqhip/optim.py:FlatAdamW (csrc/adamw.hip): the reference's AdamW update (train_rqvae.py:136-138) in one launch.  Held to torch's own
AdamW -- the foreach implementation the reference runs, and the fused one -- over several steps, state_dict interchange in both directions,
and replay from a captured hipGraph (the device-side step counter advances)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(512, 768), (256, 512), (128, 256), (32, 128), (256, 32), (3,), (1, 1), (1025,)]
    return [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]


def _run(opt_cls, steps, seed=0, **kw):
    ps = _params(seed)
    opt = opt_cls(ps, lr=1e-2, weight_decay=1e-2, **kw)
    g = torch.Generator().manual_seed(seed + 1)
    for _ in range(steps):
        for p in ps:
            p.grad = (torch.randn(p.shape, generator=g) * torch.pow(10.0, torch.randint(-6, 2, (1,), generator=g).float())).cuda()
        opt.step()
    return ps, opt


def test_flat_adamw_equals_torch_adamw():
    from rqhip.optim import FlatAdamW
    ours, _ = _run(FlatAdamW, 7)
    for kw in ({"foreach": True}, {"fused": True}):
        ref, _ = _run(torch.optim.AdamW, 7, **kw)
        for a, b in zip(ours, ref):
            assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()), kw


def test_flat_adamw_state_dict_interchange_and_skipped_gradients():
    from rqhip.optim import FlatAdamW
    ours, opt = _run(FlatAdamW, 3)
    ref, ropt = _run(torch.optim.AdamW, 3, fused=True)
    sd, rsd = opt.state_dict(), ropt.state_dict()
    assert set(sd["state"][0]) == set(rsd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == float(rsd["state"][0]["step"]) == 3.0
    # cross-load: ours continues from torch's state and the other way round, then two more steps each
    a_ps, b_ps = _params(0), _params(0)
    a, b = FlatAdamW(a_ps, lr=1e-2, weight_decay=1e-2), torch.optim.AdamW(b_ps, lr=1e-2, weight_decay=1e-2, fused=True)
    with torch.no_grad():
        for p, q, r in zip(a_ps, b_ps, ref):
            p.copy_(r)
            q.copy_(r)
    a.load_state_dict(rsd)
    b.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    for _ in range(2):
        for p, q in zip(a_ps, b_ps):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, q.grad = gr, gr.clone()
        a_ps[5].grad = None                       # a parameter without a gradient is left alone (and so is its state)
        b_ps[5].grad = None
        a.step()
        b.step()
    for p, q in zip(a_ps, b_ps):
        assert (p - q).abs().max().item() <= 1e-6 * max(1.0, q.abs().max().item())
    assert float(a.state[a_ps[0]]["step"]) == 5.0


def test_flat_adamw_in_a_captured_graph_advances_its_step_counter():
    from rqhip.optim import FlatAdamW
    ps = _params(3)
    opt = FlatAdamW(ps, lr=1e-2, weight_decay=0.0)
    static_g = [torch.randn_like(p) for p in ps]
    for p, g in zip(ps, static_g):
        p.grad = g
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert float(opt.state[ps[0]]["step"]) == 4.0           # one eager step + three replays (the capture itself executes nothing)
    ref = _params(3)
    ropt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.0, foreach=True)
    for _ in range(4):
        for p, g in zip(ref, static_g):
            p.grad = g
        ropt.step()
    for a, b in zip(ps, ref):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())


def test_flat_adamw_on_a_model_with_odd_sizes_through_the_flat_gradient_buffer():
    """gradients that are views of rqhip.dist.FlatGradReducer's buffer stay 16-byte aligned behind parameters whose numel is
    not a multiple of 4 (the slices are padded), and a misaligned gradient from anywhere else is copied instead of rejected."""
    from rqhip.dist import FlatGradReducer
    from rqhip.optim import FlatAdamW
    torch.manual_seed(5)
    shapes = [(7, 3), (5,), (9, 7), (2, 2)]
    ours = [torch.randn(s).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    red = FlatGradReducer(ours)
    opt, ropt = FlatAdamW(ours, lr=1e-2, weight_decay=1e-2), torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2, foreach=True)
    for step in range(3):
        gs = [torch.randn(s).cuda() for s in shapes]
        for p, v, q, g in zip(ours, red._views, ref, gs):
            v.copy_(g)
            p.grad = v                      # a view of the padded flat buffer
            q.grad = g.clone()
            assert v.data_ptr() % 16 == 0
        opt.step()
        ropt.step()
    for a, b in zip(ours, ref):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())
    # a packed buffer WITHOUT padding (somebody else's): the second gradient starts 84 bytes in
    packed = torch.randn(21 + 5, device="cuda")
    ours[0].grad, ours[1].grad = packed[:21].view(7, 3), packed[21:]
    ours[2].grad = ours[3].grad = None
    ref[0].grad, ref[1].grad = packed[:21].view(7, 3).clone(), packed[21:].clone()
    ref[2].grad = ref[3].grad = None
    opt.step()
    ropt.step()
    for a, b in zip(ours[:2], ref[:2]):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())
