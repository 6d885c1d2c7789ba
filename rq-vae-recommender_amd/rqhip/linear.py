"""Kernel selection for the bias-free Linear(+ReLU) layers of the MLPs, shared by the autograd Functions of
modules/encoder.py and the registered operators of rqhip/torch_ops.py (both must run the same kernels: the tests compare
them bit for bit).

The large layers run on the 16-bit matrix cores with fp32's accuracy (csrc/gemm_split.hip, csrc/wgrad_split.hip).  Round 4's
product arithmetic is `f16x2`: every row (column, for the weight gradients) is scaled by an exact power of two and split into
two fp16 pieces, three piece products per product.  The scales are the maxima of the rows / columns of the operand, which the
kernel that WROTE the operand emits from its epilogue (`Scales`, handed from layer to layer by modules/encoder.py:_MLPStack);
a caller that has none gets them from one `ops.maxima` pass -- same maxima, same exponents, same result bits either way.  (One exception:
when the reconstruction loss's upstream gradient is not the announced 1 / B -- `ops.recon_rescale_rows` -- rows that were scaled DOWN leave
the column maxima of that gradient too large: still a valid scale, one low-order bit of the fp16 low piece per binade of overestimate,
and not the bits a fresh maxima pass would give.  No shipped training loop takes that path: the loss is a mean.)
`use_arith("bf16x3")` keeps round 3's three-piece bf16 kernels for A/B (bench.py --mlp split6)."""
from typing import List, Optional, Tuple

import os

import torch
from torch import Tensor

from . import _lib, ops

# ---- the large activation GEMMs on the matrix cores (csrc/gemm_split.hip) -------------------------------------------------------
# Measured at 100 000 rows against the tuned library fp32 GEMM of the same layer (tools/bench_gemm_split.py, and in the
# step: profiles/r04_*): every supported forward and data gradient is faster.  Small batches are launch-bound and stay with
# the library.
_SPLIT_MIN_ROWS = 4096
_SPLIT_GEMMS = True
_ARITH = ops.F16X2
_FP32 = -1   # no split kernels at all: library fp32 GEMMs, fp32-MFMA weight gradients (round 2's step, bench.py --mlp library)
_ARITH_NAMES = {"f16x2": ops.F16X2, "bf16x3": ops.BF16X3, "fp32": _FP32}


def use_split_gemms(on: bool = True) -> bool:
    """Route the large activation GEMMs through csrc/gemm_split.hip (default) or the library (A/B: tools/ab_step.py).
    Returns the previous setting."""
    global _SPLIT_GEMMS
    before, _SPLIT_GEMMS = _SPLIT_GEMMS, bool(on)
    return before


def use_arith(name: str) -> str:
    """"f16x2" (two fp16 pieces under exact power-of-two scales, three products: the product path), "bf16x3" (round 3: three
    bf16 pieces, six products) or "fp32" (round 2: library fp32 GEMMs and the oracle-ordered fp32-MFMA weight gradients).
    Returns the previous setting's name."""
    global _ARITH
    before = arith_name()
    _ARITH = _ARITH_NAMES[name]
    return before


def arith_name() -> str:
    return next(k for k, v in _ARITH_NAMES.items() if v == _ARITH)


def f16() -> bool:
    return _ARITH == ops.F16X2


class Scales:
    """What the fp16 kernels need to know about an operand [M, N]: `rows` int32 [parts, M] (a row's largest |value| is the
    maximum over the parts, bit patterns), `cols` int32 [N]; either may be missing (None) until someone needs it."""
    __slots__ = ("rows", "cols")

    def __init__(self, rows: Optional[Tensor] = None, cols: Optional[Tensor] = None):
        self.rows, self.cols = rows, cols


# ---- maxima that come with the data ----------------------------------------------------------------------------------------------
# The largest |value| of an item's feature row is a property of the ITEM: the HBM-resident item matrix (data/processed.py:ItemData)
# computes it once per corpus and hands it out with every batch it gathers (`attach_scales`), and the stack that reads the batch takes
# it from there instead of running rqhip_maxima over the batch again (82 us of a 2.86 ms step at 100 000 x 768).  The column maxima the
# first layer's weight gradient wants are replaced by the corpus-wide ones, an upper bound for every batch (costs low-order bits of
# entries more than 2^16 below their column's corpus maximum only; same rule as a masked gradient under unmasked maxima).
def attach_scales(t: Tensor, rows: Optional[Tensor], cols: Optional[Tensor]) -> Tensor:
    """Remember on the tensor OBJECT `t` [M, N] the bit patterns of its row maxima (`rows`: int32 [1, M]) and of (upper bounds of)
    its column maxima (`cols`: int32 [N]); views / copies of `t` do not inherit them."""
    t._rq_scales = Scales(rows, cols)
    return t


def attached_scales(t: Tensor) -> Optional[Scales]:
    sc = getattr(t, "_rq_scales", None)
    if sc is None:
        return None
    ok_r = sc.rows is None or (sc.rows.dtype == torch.int32 and sc.rows.dim() == 2 and sc.rows.shape[1] == t.shape[0] and sc.rows.device == t.device)
    ok_c = sc.cols is None or (sc.cols.dtype == torch.int32 and tuple(sc.cols.shape) == (t.shape[1],) and sc.cols.device == t.device)
    return Scales(sc.rows if ok_r else None, sc.cols if ok_c else None)


_HANDOFF: List[Optional[Scales]] = [None]


def handoff_scales(sc: Optional[Scales]) -> None:
    """The scales of the tensor the NEXT `_MLPStack.apply` receives as its input (set by MLP._run right before the call, taken by the
    node's forward: an autograd Function may be handed an alias of the caller's tensor object, attributes do not survive that)."""
    _HANDOFF[0] = sc


def take_scales() -> Optional[Scales]:
    sc, _HANDOFF[0] = _HANDOFF[0], None
    return sc


# ---- zeroed int32 slices for column maxima ---------------------------------------------------------------------------------------------
# Every kernel that emits column maxima accumulates them with atomic maxima into a buffer the caller zeroed; a training step at 100 000
# rows asked for six such buffers, each a `torch.zeros` = one 4 us fill launch (28 us of a 2.7 ms step).  They are cut from a pool that
# is zeroed ONCE (64 K words = one fill per ~20 steps); a slice is handed out once and never reused, an exhausted pool is replaced (its
# slices live on as long as somebody holds them).  Under hipGraph capture a replay would meet non-zero slices: there, plain torch.zeros.
_ZERO_POOL = {}
_ZERO_POOL_WORDS = 1 << 16
_TRIMS = True


def use_step_trims(on: bool = True) -> bool:
    """A/B switch (bench.py --no-trims) for round 6's launch-overhead trims: pooled zero arenas (zeros_i32), the duplicate statistic on a
    side stream (modules/rqvae.py), the many-workgroup loss means (ops.loss_means).  Returns the previous setting."""
    global _TRIMS
    before, _TRIMS = _TRIMS, bool(on)
    return before


def trims_on() -> bool:
    return _TRIMS


def zeros_i32(n: int, device) -> Tensor:
    """A zeroed int32 [n] tensor nobody else holds (a 16-byte aligned slice of the device's pool; see above)."""
    if n <= 0:
        return torch.zeros((0,), dtype=torch.int32, device=device)
    if not _TRIMS or n > _ZERO_POOL_WORDS // 4 or torch.cuda.is_current_stream_capturing():
        return torch.zeros((n,), dtype=torch.int32, device=device)
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)     # (a pool belongs to the stream its fill ran on)
    pool = _ZERO_POOL.get(key)
    need = (n + 3) // 4 * 4
    if pool is None or pool[1] + need > _ZERO_POOL_WORDS:
        pool = [torch.zeros((_ZERO_POOL_WORDS,), dtype=torch.int32, device=device), 0]
        _ZERO_POOL[key] = pool
    out = pool[0][pool[1]:pool[1] + n]
    pool[1] += need
    return out


_GRAD_HANDOFF: List[Optional[tuple]] = [None]


def handoff_grad(g: Tensor, sc: Scales) -> None:
    """The NEXT `_MLPStack.backward` that receives `g` (same storage, same shape) as its upstream gradient may take it as already masked by
    its last layer's ReLU, with these maxima (set by the backward of modules/rqvae.py's seam node, whose epilogue did both)."""
    _GRAD_HANDOFF[0] = (g.data_ptr(), tuple(g.shape), sc)


def take_grad_handoff(g: Tensor) -> Optional[Scales]:
    slot, _GRAD_HANDOFF[0] = _GRAD_HANDOFF[0], None     # one shot: whoever asks next clears it, match or not
    if slot is not None and slot[0] == g.data_ptr() and slot[1] == tuple(g.shape):
        return slot[2]
    return None


def ensure_scales(a: Tensor, sc: Optional[Scales], rows: bool, cols: bool) -> Scales:
    """`sc` with the requested maxima present: what is missing is computed by ONE pass over `a` (rqhip_maxima)."""
    sc = sc if sc is not None else Scales()
    need_r, need_c = rows and sc.rows is None, cols and sc.cols is None
    if need_r or need_c:
        r, c, _ = ops.maxima(a, rows=need_r, cols=need_c)
        if need_r:
            sc.rows = r
        if need_c:
            sc.cols = c
    return sc


def split_ok(x: Tensor, n_cols: int, n_red: int, forward_relu: bool = False) -> bool:
    """Does this GEMM (x [M, n_red] against a weight image of n_cols columns) take the split kernel?"""
    del forward_relu   # (round 3 excluded the first encoder layer's forward here)
    return bool(_SPLIT_GEMMS and _ARITH != _FP32 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and x.shape[0] >= _SPLIT_MIN_ROWS and x.is_contiguous() and _wide(n_cols)
                and ops.gemm_split_supported(n_cols, n_red))


def _wide(n_cols: int) -> bool:
    """Layers of 128 (mod 128) output columns: 256-column tiles (the product kernel) or, for 128 (mod 256) columns, the staged
    loop's 128-column tile.  Round 3 left the latter to the library (HBM-bound shapes, measured level: 100 000 x 256 -> 128
    forward 58 us vs 67, its data gradient 59 vs 56); with the f16x2 arithmetic, maxima from the epilogues and the transposing
    epilogue the step is 2 % faster with them on the split kernel (3.06 -> 2.99 ms, bench.py --no-narrow for the A/B)."""
    return n_cols % 256 == 0 or (_NARROW and n_cols % 128 == 0)


_NARROW = True


def use_narrow_tiles(on: bool = True) -> bool:
    """A/B switch (bench.py --no-narrow): layers of 128 (mod 256) output columns through the split kernel's 128-column tile as well."""
    global _NARROW
    before, _NARROW = _NARROW, bool(on)
    return before


def split_shape_ok(rows: int, n_cols: int, n_red: int) -> bool:
    """`split_ok` for a contiguous fp32 ROCm operand of `rows` rows that does not exist yet."""
    return bool(_SPLIT_GEMMS and _ARITH != _FP32 and rows >= _SPLIT_MIN_ROWS and _wide(n_cols)
                and ops.gemm_split_supported(n_cols, n_red))


def wgrad_f16_ok(n_out: int, n_in: int, rows: int = _SPLIT_MIN_ROWS) -> bool:
    """Does the weight gradient of an [n_out, n_in] layer over `rows` batch rows run on the fp16 split kernel (and therefore
    want column maxima)?  The shapes csrc/wgrad_split.hip tiles: both multiples of 128, one of them of 256 -- and batches of
    4096 rows and more: below that a step is launch-bound, and the maxima passes / zeroed arenas the scales need are launches
    (batch 640: 71 device activities per step with them, 1.55 ms; the fused-mask kernels of csrc/wgrad.hip need none)."""
    return bool(f16() and rows >= _SPLIT_MIN_ROWS and n_out % 128 == 0 and n_in % 128 == 0 and (n_out % 256 == 0 or n_in % 256 == 0))


_WGRAD_BATCH = True


def use_wgrad_batch(on: bool = True) -> bool:
    """A/B (round 6): at split-kernel batch sizes the weight gradients of an MLP stack's 256 x 256-tiled layers in ONE launch
    (rqhip_linear_wgrad_f16_batch: half the partial-block traffic of two launches; default) or one launch per layer.  Returns the
    previous setting."""
    global _WGRAD_BATCH
    before, _WGRAD_BATCH = _WGRAD_BATCH, bool(on)
    return before


# ---- ... and of TWO stacks in one launch: the decoder's wait for the encoder's (single rank, eager, gradients in a flat buffer) ----------
# The decoder's backward runs first; its 256 x 256-tiled weight gradients (with the encoder's: 16 tiles x 16 row ranges instead of twice
# 8 x 32) are launched at the end of the encoder stack's backward, or by a callback the autograd engine runs when the backward pass ends
# (an encoder without gradients, a stack used on its own).  What the decoder's node returns for those weights meanwhile is the parameter's
# slice of the flat gradient buffer (rqhip/dist.py:claim_grad_sink), which the launch fills before anything on the stream reads it.  NOT
# with several ranks: the decoder's gradients go on the wire under the encoder's backward there (FlatGradReducer.boundary_hook).
_XSTACK_ON = True
_XSTACK: List[tuple] = []            # (w, g, x, g_cols, x_cols, sink) waiting for a later stack of the same backward pass
_DEFER_NEXT = [False]


def use_wgrad_cross_stack(on: bool = True) -> bool:
    """A/B (round 6): the decoder's batched weight gradients wait for the encoder's launch (default) or are launched per stack."""
    global _XSTACK_ON
    before, _XSTACK_ON = _XSTACK_ON, bool(on)
    return before


def mark_next_stack_defers(on: bool = True) -> None:
    """modules/encoder.py:MLP._run for a module tagged `_defer_wgrads` (RqVae's decoder): the stack node created next may hand its
    batched weight gradients to a later node's launch."""
    _DEFER_NEXT[0] = bool(on)


def take_defer_flag() -> bool:
    f, _DEFER_NEXT[0] = _DEFER_NEXT[0], False
    return f


def xstack_ok() -> bool:
    from . import dist as _dist
    return bool(_XSTACK_ON and _WGRAD_BATCH and _dist.world_size() == 1 and not torch.cuda.is_current_stream_capturing())


def _launch_wgrads(jobs: List[tuple]) -> None:
    """jobs = [(w, g, x, g_cols, x_cols, sink), ...]: batched launches where the plan allows (layers tiled 256 x 256 together, layers tiled
    128 x 256 / 256 x 128 together; at most 4 per launch), else per layer."""
    full = [j for j in jobs if j[0].shape[0] % 256 == 0 and j[0].shape[1] % 256 == 0]
    half = [j for j in jobs if not (j[0].shape[0] % 256 == 0 and j[0].shape[1] % 256 == 0)]
    for group in (full, half):
        _launch_wgrad_group(group)


def _launch_wgrad_group(jobs: List[tuple]) -> None:
    while jobs:
        take, jobs = jobs[:4], jobs[4:]
        M = take[0][1].shape[0]
        if len(take) >= 2 and ops.linear_wgrad_f16_batch_ranges(M, [tuple(w.shape) for w, *_ in take]) >= 1:
            ops.linear_wgrad_f16_batch([(g, x, gc, xc) for _, g, x, gc, xc, _ in take], outs=[sk for *_, sk in take])
        else:
            for w, g, x, gc, xc, sk in take:
                weight_grad(g, None, x, w, out=sk, want_masked=False, g_scales=Scales(None, gc), x_scales=Scales(None, xc), premasked=True)


_XSTACK_STREAM: List = [None]      # the stream the waiting stack's backward ran on


def xstack_flush() -> None:
    """Launch whatever still waits: the autograd engine's end-of-backward callback.  It runs on the thread that called backward(), whose
    current stream need not be the one the backward nodes ran on -- the launch goes to THAT stream (recorded by xstack_push)."""
    if _XSTACK:
        jobs, _XSTACK[:] = list(_XSTACK), []
        st = _XSTACK_STREAM[0]
        if st is not None and st != torch.cuda.current_stream(st.device):
            with torch.cuda.stream(st):
                _launch_wgrads(jobs)
        else:
            _launch_wgrads(jobs)


def xstack_push(jobs: List[tuple]) -> None:
    if not _XSTACK:
        torch.autograd.Variable._execution_engine.queue_callback(xstack_flush)
        _XSTACK_STREAM[0] = torch.cuda.current_stream()
    _XSTACK.extend(jobs)


def xstack_take() -> List[tuple]:
    jobs, _XSTACK[:] = list(_XSTACK), []
    return jobs


# ---- the same for the job-table launch of the reference's batch sizes (< 4096 rows, csrc/wgrad_jobs.hip): one launch for BOTH stacks' layers --------
# (also inside a hipGraph capture: the engine's end-of-backward callback runs inside the captured region like everything else of the step)
_XSMALL: List[tuple] = []            # (g, x, sink)


def xsmall_ok() -> bool:
    from . import dist as _dist
    return bool(_XSTACK_ON and _WGRAD_JOBS and _dist.world_size() == 1)


def _launch_small(jobs: List[tuple]) -> None:
    while jobs:
        take, jobs = jobs[:ops.WGRAD_JOBS_MAX], jobs[ops.WGRAD_JOBS_MAX:]
        ops.linear_wgrad_jobs([(g, x) for g, x, _ in take], outs=[sk for *_, sk in take])


def xsmall_flush() -> None:
    if _XSMALL:
        jobs, _XSMALL[:] = list(_XSMALL), []
        st = _XSTACK_STREAM[0]
        if st is not None and st != torch.cuda.current_stream(st.device):
            with torch.cuda.stream(st):
                _launch_small(jobs)
        else:
            _launch_small(jobs)


def xsmall_push(jobs: List[tuple]) -> None:
    if not _XSMALL:
        torch.autograd.Variable._execution_engine.queue_callback(xsmall_flush)
        _XSTACK_STREAM[0] = torch.cuda.current_stream()
    _XSMALL.extend(jobs)


def xsmall_take() -> List[tuple]:
    jobs, _XSMALL[:] = list(_XSMALL), []
    return jobs


def wgrad_batch_shape_ok(n_out: int, n_in: int, rows: int) -> bool:
    """May this layer's weight gradient wait for the stack's batched launch?"""
    return bool(_WGRAD_BATCH and wgrad_f16_ok(n_out, n_in, rows))      # (tiled 256 x 256, 128 x 256 or 256 x 128: ops.linear_wgrad_f16_batch)


_WGRAD_JOBS = True


def use_wgrad_jobs(on: bool = True) -> bool:
    """Batches below the split kernels' 4096 rows -- the sizes the reference's gin files train with -- take the job-table
    weight-gradient kernel (csrc/wgrad_jobs.hip: every layer of an MLP stack in one launch, no row ranges, no reduction launches;
    default on).  Off: round 4's per-layer kernels (A/B: tools/bench_small_batch.py --no-jobs).  Returns the previous setting."""
    global _WGRAD_JOBS
    before, _WGRAD_JOBS = _WGRAD_JOBS, bool(on)
    return before


def wgrad_jobs_ok(rows: int, shapes) -> bool:
    """Do the weight gradients of layers `shapes` = [(n_out, n_in), ...] over `rows` batch rows run as ONE job-table launch?"""
    shapes = list(shapes)
    return bool(_WGRAD_JOBS and _ARITH != _FP32 and 0 < rows < _SPLIT_MIN_ROWS and 0 < len(shapes) <= ops.WGRAD_JOBS_MAX
                and all(ops.linear_wgrad_jobs_supported(n, k) for n, k in shapes))


_WIDE_TILES = False
_WIDE_MIN_RED = 512


def use_wide_tiles(on: bool = True) -> bool:
    """A/B arm, default OFF (round 6): layers of 512 (mod 512) output columns with a reduction of 512 and more on
    full-output-width tiles (128 x 512, 8 waves, ONE workgroup per CU: a strip of A is fetched and split once per row tile) instead of
    two 128 x 256 tiles of two workgroups per CU.  Same result bits; HBM traffic and VALU work per matrix instruction fall as intended
    and the kernels get 4-6 % SLOWER inside the step (768 -> 512 forward 258 -> 273 us, its data gradient 269 -> 280 us:
    profiles/r06_gemm_wide_ab.txt) -- eight waves behind one barrier idle together, two independent workgroups do not.
    bench.py --wide-tiles, tools/gemm_wide_ab.py.  Returns the previous setting."""
    global _WIDE_TILES
    before, _WIDE_TILES = _WIDE_TILES, bool(on)
    return before


_TINY_TILES = False
_ROWS_WITH_COLS = True     # (the note in `gemm` below; A/B on one box: 2.5385 vs 2.5456 ms, profiles/r06_step_ab.txt)
_STATIC_TILES = True


def use_static_tiles(on: bool = True) -> bool:
    """A/B (round 6): the product GEMM's tiles dealt statically (workgroup b: tiles b, b + grid, ...; default) or handed out by the
    atomic dispenser of rounds 4-5.  Returns the previous setting."""
    global _STATIC_TILES
    before, _STATIC_TILES = _STATIC_TILES, bool(on)
    return before


def use_tiny_tiles(on: bool = True) -> bool:
    """A/B (round 6, neutral in the step, off by default): the product GEMM's leftover behind the whole rounds of 128-row tiles as 32-row
    tiles where the launch plan prices them cheaper, instead of round 5's plan (64-row leftover tiles / all big).  Returns the previous
    setting."""
    global _TINY_TILES
    before, _TINY_TILES = _TINY_TILES, bool(on)
    return before


def _tile_code(n_cols: int, n_red: int, epilogue: int) -> int:
    """rqhip_gemm_args.tile_rows for a layer: -5 selects the 128 x 512 tile (csrc/gemm_split.hip), -9 allows 32-row leftover tiles, -12 the atomic tile dispenser, 0 the default."""
    if _WIDE_TILES and f16() and n_cols % 512 == 0 and n_red >= _WIDE_MIN_RED and epilogue != _lib.EPI_RECON:
        return -5
    if not _STATIC_TILES:
        return -12
    return -9 if _TINY_TILES else 0


# ---- the 32-wide layers either side of the quantiser (the RQ <-> MLP seam) ------------------------------------------------------------
# The encoder's last Linear (128 -> 32) and the decoder's first (32 -> 128), forward and data gradient, run on csrc/rq_forward.hip's seam
# kernel (rqhip_rq_seam): every output is ONE fp32 FMA chain over the input features on the fp32 matrix pipe -- no library call, the ReLU /
# ReLU backward and the maxima the neighbouring split kernels scale by in the same launch, and the same bits in the fused launch
# RqVae.forward uses (modules/rqvae.py), which is this kernel with the quantisation levels switched on.  Batches of 4096 rows and more.
_CHAIN = True
CHAIN_D, CHAIN_H = 32, ops.SEAM_H


def use_chain_gemms(on: bool = True) -> bool:
    """The 128 <-> 32 layers on the seam kernel (default) or on the library / split kernels as in round 5 (A/B: bench.py --no-seam).
    Returns the previous setting."""
    global _CHAIN
    before, _CHAIN = _CHAIN, bool(on)
    return before


def chain_shape(n_out: int, n_in: int, rows: int = _SPLIT_MIN_ROWS) -> int:
    """0: not a seam layer; 1: 128 -> 32 (rows enter the input GEMM); 2: 32 -> 128 (rows leave through the output GEMM).
    Batches of 4096 rows and more, like the split kernels: every output of these GEMMs is a 64- or 16-deep chain of dependent fp32
    matrix instructions over 32 rows at a time -- at the reference's batch 640 that is 20 such chains on 20 SIMDs, 9.3 us per
    launch against 4.6-5.9 us for the library's 16 x 16 tiles (profiles/r06_seam.txt): small batches take rqhip_linear_small (below),
    which splits every reduction over four waves per tile."""
    if not _CHAIN or rows < _SPLIT_MIN_ROWS:
        return 0
    return 1 if (n_out, n_in) == (CHAIN_D, CHAIN_H) else 2 if (n_out, n_in) == (CHAIN_H, CHAIN_D) else 0


def chain_kind(x: Tensor, n_out: int, n_in: int) -> int:
    """chain_shape for an operand that exists: a 2-D fp32 ROCm tensor of n_in columns with at least one row, 16-byte aligned rows."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and x.shape[1] == n_in
            and x.data_ptr() % 16 == 0):
        return 0
    return chain_shape(n_out, n_in, x.shape[0])


def _chain_scales(r, col_out) -> "Scales":
    return Scales(r.out_row_max if r.out_row_max is not None else None, col_out)


def chain_forward(x: Tensor, w: Tensor, relu: bool, want_rows: bool = False, col_out: Optional[Tensor] = None):
    """(y, Scales of y) of y = [relu](x w^T) for a seam layer (chain_kind != 0; a ReLU only behind the 32 -> 128 layer)."""
    kind = chain_kind(x, w.shape[0], w.shape[1])
    x = x if x.is_contiguous() else x.contiguous()
    if kind == 1:
        assert not relu
        return ops.rq_seam(h=x, w_in=w.detach()).res0, Scales()
    r = ops.rq_seam(res0=x, w_out=w.detach(), epilogue=_lib.EPI_RELU if relu else _lib.EPI_STORE, want_row_max=want_rows and f16(),
                    col_max_out=col_out if f16() else None)
    return r.out, _chain_scales(r, col_out if f16() else None)


def chain_input_grad(g: Tensor, w: Tensor, *, g_mask: Optional[Tensor] = None, out_mask: Optional[Tensor] = None, want_rows: bool = False,
                     col_out: Optional[Tensor] = None):
    """(gx, Scales of gx) of gx = g' w for a seam layer with weight w [n_out, n_in]; g' = g where g_mask > 0 (the layer's own ReLU
    backward, applied on load: 32 -> 128 layers only); gx is kept where out_mask > 0 (the ReLU backward of the layer below, in the
    epilogue: 128 -> 32 layers only)."""
    kind = chain_kind(g, w.shape[1], w.shape[0])      # the data gradient maps n_out -> n_in
    g = g if g.is_contiguous() else g.contiguous()
    if kind == 1:       # layer 32 -> 128, w [128, 32]: gx [B, 32] = g' [B, 128] . w
        assert out_mask is None
        return ops.rq_seam(h=g, h_mask=g_mask, w_in=w.detach(), w_in_transposed=True).res0, Scales()
    assert kind == 2 and g_mask is None   # layer 128 -> 32, w [32, 128]: gx [B, 128] = g [B, 32] . w
    r = ops.rq_seam(res0=g, w_out=w.detach(), w_out_transposed=True, epilogue=_lib.EPI_MASK if out_mask is not None else _lib.EPI_STORE,
                    out_mask=out_mask, want_row_max=want_rows and f16(), col_max_out=col_out if f16() else None)
    return r.out, _chain_scales(r, col_out if f16() else None)


# ---- batches below 4096 rows: the layers' forward and data gradient on csrc/mlp_small.hip (rqhip_linear_small) -----------------------
# Exact fp32 on the fp32 matrix instruction, the ReLU / the ReLU backward of the layer below in the epilogue: at the reference's batch
# sizes (640 / 64 rows) a training step has no library GEMM and no threshold_backward launch left (rounds 1-5: 16 + 6 of its 39 launches).
_SMALL = True


def use_small_kernels(on: bool = True) -> bool:
    """rqhip_linear_small below 4096 rows (default) or the library GEMMs as in round 5 (A/B: tools/bench_small_batch.py --no-small).
    Returns the previous setting."""
    global _SMALL
    before, _SMALL = _SMALL, bool(on)
    return before


def small_shape_ok(rows: int, n_out: int, n_red: int) -> bool:
    """Does out [rows, n_out] = a [rows, n_red] . B run on rqhip_linear_small?  (Not in the strict-fp32 arm, which means "library".)"""
    return bool(_SMALL and _ARITH != _FP32 and 0 < rows < _SPLIT_MIN_ROWS and n_out % 32 == 0 and n_red % 32 == 0)


def small_ok(a: Tensor, n_out: int, n_red: int) -> bool:
    return bool(a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.shape[1] == n_red and a.data_ptr() % 16 == 0
                and small_shape_ok(a.shape[0], n_out, n_red))


def small_forward(x: Tensor, w: Tensor, relu: bool) -> Tensor:
    """[relu](x w^T), small_ok(x, *w.shape)."""
    return ops.linear_small(x, w.detach(), epilogue=_lib.EPI_RELU if relu else _lib.EPI_STORE)


def small_input_grad(g: Tensor, w: Tensor, below: Optional[Tensor] = None) -> Tensor:
    """g w for w [n_out, n_in], kept where `below` [M, n_in] > 0 (the ReLU backward of the layer below) when given."""
    return ops.linear_small(g, w.detach(), w_kn=True, epilogue=_lib.EPI_MASK if below is not None else _lib.EPI_STORE, aux=below)


def images(jobs: List[Tuple[Tensor, bool]]) -> List[Tensor]:
    """The weight images of `jobs` = [(w, transpose), ...] in the current arithmetic, one launch.  Rebuilt at EVERY forward:
    a first version cached an image per `w._version` -- and trained on stale weights: the fused AdamW update (and any
    `w.data` write) does not bump the version counter.  The images of an MLP's forward and data-gradient GEMMs are built
    together when its forward starts (weights cannot change between a forward and its backward); inside a hipGraph capture
    the build is part of the graph, as it has to be."""
    return ops.weight_images([(w.detach(), tr) for w, tr in jobs], _ARITH)


def planes(w: Tensor, transpose: bool) -> Tensor:
    return images([(w, transpose)])[0]


def gemm(a: Tensor, image: Tensor, n_cols: int, *, epilogue: int = _lib.EPI_STORE, aux: Optional[Tensor] = None,
         row_scale: float = 0.0, a_scales: Optional[Scales] = None, want_rows: bool = False,
         col_out: Optional[Tensor] = None):
    """One split GEMM in the current arithmetic.  Returns (C, loss_rows, Scales of C): the row maxima of `a` come from
    `a_scales` or from a pass; the Scales of C hold the row maxima when `want_rows` and `col_out` (a zeroed int32 [n_cols]
    slice that receives the column maxima) when given -- both only in the f16x2 arithmetic (the bf16 path needs none)."""
    rows_in = ensure_scales(a, a_scales, True, False).rows if f16() else None
    # (a launch that emits column maxima also emits row maxima, wanted or not: the kernel's straight-line epilogue is the form with both --
    # csrc/gemm_split.hip:gs_epilogue -- and the general loop costs such a launch ~10 %; the extra [column tiles, M] words are dropped)
    c, loss_rows, crm = ops.gemm_split_ex(a, image, n_cols, arith=_ARITH, epilogue=epilogue, aux=aux, row_scale=row_scale,
                                          a_row_max=rows_in, want_row_max=(want_rows or (col_out is not None and _ROWS_WITH_COLS)) and f16(),
                                          col_max_out=col_out if f16() else None, tile_rows=_tile_code(n_cols, a.shape[1], epilogue))
    return c, loss_rows, Scales(crm if want_rows else None, col_out if f16() else None)


def input_grad(g: Tensor, w: Tensor, *, g_scales: Optional[Scales] = None, image: Optional[Tensor] = None) -> Tensor:
    """g [M, N] . w [N, K]: the split kernel with the image of w^T where it applies, else the library GEMM."""
    g = g if g.is_contiguous() else g.contiguous()
    if chain_kind(g, w.shape[1], w.shape[0]):
        return chain_input_grad(g, w)[0]
    if split_ok(g, w.shape[1], w.shape[0], False):
        return gemm(g, image if image is not None else planes(w, True), w.shape[1], a_scales=g_scales)[0]
    if small_ok(g, w.shape[1], w.shape[0]) and w.is_contiguous() and w.data_ptr() % 16 == 0:
        return small_input_grad(g, w)
    return g.mm(w)


def forward(x: Tensor, w: Tensor, relu: bool, zero_bias: Tensor = None) -> Tensor:
    """relu(x w^T) or x w^T for 2-D fp32 ROCm tensors: the split kernel where it applies, else the library GEMM (with
    the ReLU in the hipBLASLt epilogue)."""
    kind = chain_kind(x, w.shape[0], w.shape[1])
    if kind == 2 or (kind == 1 and not relu):
        return chain_forward(x, w, relu)[0]
    if split_ok(x, w.shape[0], w.shape[1], relu):
        return gemm(x, planes(w, False), w.shape[0], epilogue=_lib.EPI_RELU if relu else _lib.EPI_STORE)[0]
    if small_ok(x, w.shape[0], w.shape[1]) and x.is_contiguous() and w.is_contiguous() and w.data_ptr() % 16 == 0:
        return small_forward(x, w, relu)
    return library_forward(x, w, relu, zero_bias)


def library_forward(x: Tensor, w: Tensor, relu: bool, zero_bias: Tensor = None) -> Tensor:
    if relu:
        zb = zero_bias if zero_bias is not None else x.new_zeros((w.shape[0],))
        return torch._addmm_activation(zb, x, w.t())
    return x.mm(w.t())


def hip_wgrad_ok(g: Tensor, w: Tensor) -> bool:
    return bool(g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.shape[0] > 0
                and ops.linear_wgrad_supported(w.shape[0], w.shape[1]))


def weight_grad(g: Tensor, y: Optional[Tensor], x: Tensor, w: Tensor, *, out: Optional[Tensor] = None,
                want_masked: bool = True, g_scales: Optional[Scales] = None, x_scales: Optional[Scales] = None,
                premasked: bool = False):
    """(dW, g_pre, Scales of g_pre) of y = relu(x w^T) (y given) or y = x w^T (y None).  dW = g_pre^T x with
    g_pre = g where y > 0 else 0; `premasked`: g already is g_pre (a data gradient whose epilogue applied this layer's ReLU
    backward).  f16x2 arithmetic on the shapes csrc/wgrad_split.hip tiles: column scales of g_pre and x from `g_scales` /
    `x_scales` or from passes (for an unmasked g with y, ONE pass masks, writes g_pre and takes its maxima); otherwise the
    round-3 kernels (mask fused into the weight-gradient kernel).  Falls back to library GEMMs for shapes no kernel tiles."""
    if not hip_wgrad_ok(g, w):
        gp = g if (y is None or premasked) else torch.ops.aten.threshold_backward(g, y, 0.0)
        gw = torch.mm(gp.t(), x, out=out) if out is not None else gp.t().mm(x)
        return gw, gp, None
    if premasked:
        y = None
    if g.is_cuda and wgrad_jobs_ok(g.shape[0], [tuple(w.shape)]):     # small batches: the job-table kernel with one job
        gp = g if y is None else torch.ops.aten.threshold_backward(g, y, 0.0)
        gw = ops.linear_wgrad_jobs([(gp, x)], outs=[out] if out is not None else None)[0]
        return gw, gp, (g_scales if y is None else None)
    if wgrad_f16_ok(w.shape[0], w.shape[1], g.shape[0]):
        if y is not None:   # mask + maxima in one pass; the weight-gradient kernel then runs without a mask
            r, c, g = ops.maxima(g, y, rows=want_masked, cols=True, write_masked=True)
            g_scales = Scales(r, c)
        else:
            g_scales = ensure_scales(g, g_scales, False, True)
        x_scales = ensure_scales(x, x_scales, False, True)
        gw, _ = ops.linear_wgrad(g, None, x, out=out, g_col_max=g_scales.cols, x_col_max=x_scales.cols)
        return gw, g, g_scales
    gw, gp = ops.linear_wgrad(g, y, x, want_masked=want_masked, out=out, exact_fp32=_ARITH == _FP32)
    return gw, gp, (g_scales if y is None else None)


def backward(gy: Tensor, y: Optional[Tensor], x: Tensor, w: Tensor, need_x: bool, need_w: bool, sink: Optional[Tensor] = None):
    """(gx, gw) of one layer, y = relu(x w^T) (y given) or x w^T (y None): the per-layer backward both the autograd Functions
    and the registered operators run."""
    gy = gy if gy.is_contiguous() else gy.contiguous()
    g, gw, gs = gy, None, None
    if need_w:
        gw, g, gs = weight_grad(gy, y, x, w, out=sink, want_masked=need_x)
    elif y is not None and need_x:
        g = torch.ops.aten.threshold_backward(gy, y, 0.0)      # gy where y > 0 (what autograd does for relu)
    gx = input_grad(g, w, g_scales=gs) if need_x else None
    return gx, gw
