"""Fused memory-bound operators of the blocks (csrc/fused_ops.hip) with their autograd glue.

Each helper computes exactly what the reference's module code computes (cited at the call sites);
it takes the fused HIP path when the tensors are what bf16 autocast produces on the GPU (fp32
residual stream, bf16 branch outputs) and otherwise evaluates the same expression with torch ops
(fp32 runs, CPU host-logic tests).
"""
import os

import torch
import torch.nn.functional as F

import _vah

ENABLED = {'pair_core': True, 'layer_norm': True, 'residual': True, 'residual_ln': True, 'dwconv': True, 'linear': True, 'bn_tail': True,
           'bn_relu': True, 'bias_fold': True, 'keep_feat': True, 'maps': True, 'maps_in': True, 'linear_pair': True, 'maxpool': True, 'conv1x1': True, 'ln_dual': True, 'wgrad_fin': True, 'spm_nhwc': True, 'up_gemm': True, 'patch_gemm': True, 'wgrad_overlap': True, 'drop_pool': True}
for _k in os.environ.get('VAH_FUSED_DISABLE', '').split(','):      # e.g. VAH_FUSED_DISABLE=residual_ln,bn_tail (A/B runs)
    if _k:
        ENABLED[_k.strip()] = False


def _stream(t):
    return _vah.raw_stream(t.device)


def _scratch(K, device):
    """Partial-sum scratch for the column reductions (stream-ordered: allocated per call from
    torch's caching allocator, so concurrent streams never share it)."""
    return torch.empty(_vah.lib.vah_reduce_ws_floats(K), dtype=torch.float32, device=device)


def _bf16_autocast():
    return torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16


class _LayerNormBF16(torch.autograd.Function):
    """LayerNorm of the fp32 residual stream with bf16 output.  With ``keep`` the stream itself is
    returned next to the normalised copy, so the gradient of the residual branch and the LayerNorm
    gradient meet in ONE backward call and are summed inside the kernel (otherwise autograd adds
    them with a separate fp32 add per LayerNorm)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, keep):
        C = x.shape[-1]
        xc = x.contiguous()
        x2 = xc.view(-1, C)
        rows = x2.shape[0]
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        w, b = weight.contiguous(), bias.contiguous()
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_layernorm_fwd_f32_bf16(
                x2.data_ptr(), w.data_ptr(), b.data_ptr(), rows, C, float(eps), y.data_ptr(),
                mean.data_ptr(), rstd.data_ptr(), _stream(x)), 'layernorm_fwd')
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        if keep:
            return xc, y
        return y

    @staticmethod
    def backward(ctx, *grads):
        gres, g = grads if len(grads) == 2 else (None, grads[0])
        x2, w, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        if g is None:                      # the normalised copy was not used
            return (gres, None, None, None, None)
        g = g.contiguous().to(torch.bfloat16)
        if gres is not None:
            gres = gres.contiguous().float()
        dx = torch.empty_like(x2)
        dwb = torch.empty(2, C, dtype=torch.float32, device=x2.device)
        ws = _scratch(2 * C, x2.device)
        with _vah.on(x2.device):
            _vah.check(_vah.lib.vah_layernorm_bwd_f32_bf16(
                x2.data_ptr(), g.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                gres.data_ptr() if gres is not None else None, rows, C,
                dx.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), _stream(x2)),
                'layernorm_bwd')
        return dx.view(ctx.shape), dwb[0], dwb[1], None, None


def _ln_fusable(norm, x):
    return (ENABLED['layer_norm'] and x.is_cuda and x.dtype == torch.float32 and _bf16_autocast()
            and isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and x.numel() > 0)


class _LayerNormDualBF16(torch.autograd.Function):
    """(x, norm_a(x), norm_b(x)) for two LayerNorms of the same fp32 tensor (equal eps): statistics
    shared, x read once; the backward sums the residual-branch gradient and both LayerNorm gradients in
    one pass."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, eps):
        C = x.shape[-1]
        xc = x.contiguous()
        x2 = xc.view(-1, C)
        rows = x2.shape[0]
        ya = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        yb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        wa, ba, wb, bb = (t.contiguous() for t in (wa, ba, wb, bb))
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_layernorm_dual_fwd(
                x2.data_ptr(), wa.data_ptr(), ba.data_ptr(), wb.data_ptr(), bb.data_ptr(), rows, C, float(eps),
                ya.data_ptr(), yb.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream(x)), 'layernorm_dual_fwd')
        ctx.save_for_backward(x2, wa, wb, mean, rstd)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        return xc, ya, yb

    @staticmethod
    def backward(ctx, gres, ga, gb):
        x2, wa, wb, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        if ga is None and gb is None:
            return (gres, None, None, None, None, None)
        ga = ga.contiguous().to(torch.bfloat16) if ga is not None else None
        gb = gb.contiguous().to(torch.bfloat16) if gb is not None else None
        if gres is not None:
            gres = gres.contiguous().float()
        dx = torch.empty_like(x2)
        dp = torch.empty(4, C, dtype=torch.float32, device=x2.device)
        ws = _scratch(2 * C, x2.device)
        with _vah.on(x2.device):
            _vah.check(_vah.lib.vah_layernorm_dual_bwd(
                x2.data_ptr(), ga.data_ptr() if ga is not None else None, gb.data_ptr() if gb is not None else None,
                wa.data_ptr(), wb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                gres.data_ptr() if gres is not None else None, rows, C, dx.data_ptr(), dp.data_ptr(), ws.data_ptr(),
                _stream(x2)), 'layernorm_dual_bwd')
        return dx.view(ctx.shape), dp[0], dp[1], dp[2], dp[3], None


def layer_norm_dual_ok(norm_a, norm_b, x):
    return (ENABLED['ln_dual'] and _ln_fusable(norm_a, x) and _ln_fusable(norm_b, x) and norm_a.eps == norm_b.eps
            and x.shape[-1] <= 1024)


def layer_norm_dual_keep(norm_a, norm_b, x):
    """``(x, norm_a(x), norm_b(x))``; use the returned x downstream (see layer_norm_keep)."""
    if layer_norm_dual_ok(norm_a, norm_b, x):
        return _LayerNormDualBF16.apply(x, norm_a.weight, norm_a.bias, norm_b.weight, norm_b.bias, norm_a.eps)
    x, ya = layer_norm_keep(norm_a, x, fan_out=True)
    x, yb = layer_norm_keep(norm_b, x)
    return x, ya, yb


def layer_norm(norm, x):
    """``norm(x)`` for an nn.LayerNorm; bf16 output when the consumer is a bf16 GEMM (autocast)."""
    if _ln_fusable(norm, x):
        return _LayerNormBF16.apply(x, norm.weight, norm.bias, norm.eps, False)
    return norm(x)


def layer_norm_keep(norm, x, fan_out=False):
    """``(x, norm(x))`` for the pattern ``x + branch(norm(x))``: use the returned ``x`` for the
    residual update so both gradients of ``x`` are summed inside the LayerNorm backward kernel.
    ``fan_out``: the kept tensor is not a residual of this sub-block but goes on to other consumers
    (switchable for A/B runs: VAH_FUSED_DISABLE=keep_feat)."""
    if fan_out and not ENABLED['keep_feat']:
        return x, layer_norm(norm, x)
    if _ln_fusable(norm, x):
        return _LayerNormBF16.apply(x, norm.weight, norm.bias, norm.eps, True)
    return x, norm(x)


# ---------------------------------------------------------------------------------------
# nn.Linear under bf16 autocast
# ---------------------------------------------------------------------------------------
class _Bf16Copies:
    """bf16 working copies of fp32 parameters.  autocast makes the same copies, but one tiny cast
    kernel per parameter per use (and one more per gradient on the way back).  Here a model's forward
    opens an EPOCH (``begin_forward``): ONE multi-tensor cast writes fresh copies of all its Linear
    parameters into one new flat buffer, and those copies serve every use until the forward ends
    (``forward_epoch`` is the context manager around both).  Outside an epoch every use makes its own cast.

    Nothing is keyed on ``Tensor._version``: in-place writes through ``.data`` (legacy optimizers,
    EMA / fp16 hooks, ``weight.data.normal_()``) do not bump it, and a cached copy would silently go
    stale.  Copies are never rewritten in place either, so a backward that saved one still sees the
    values of its own forward whatever ran in between."""

    def __init__(self):
        self.entries = {}          # id(param) -> (weakref(param), copy, epoch)
        self.epoch = 0             # even: no forward open; odd: the open forward's epoch

    def get(self, p):
        e = self.entries.get(id(p))
        if e is not None and e[2] == self.epoch and (self.epoch & 1) and e[0]() is p and e[1].device == p.device:
            return e[1]
        return p.detach().to(torch.bfloat16)

    def begin(self, params):
        import weakref
        self.epoch += 1 if (self.epoch & 1) == 0 else 2          # a nested / aborted forward just opens a new epoch
        if not params:
            return
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.bfloat16, device=params[0].device)
        views = [v.view(p.shape) for v, p in zip(flat.split([p.numel() for p in params]), params)]
        torch._foreach_copy_(views, [p.detach() for p in params])
        if len(self.entries) > 4096:               # drop entries of collected parameters
            self.entries = {k: v for k, v in self.entries.items() if v[0]() is not None}
        for p, v in zip(params, views):
            self.entries[id(p)] = (weakref.ref(p), v, self.epoch)

    def end(self):
        if self.epoch & 1:
            self.epoch += 1


BF16_COPIES = _Bf16Copies()


class _DropPool:
    """The per-sample drop-path scales of ONE forward from ONE random draw.  A ViT-Adapter-B step has 28 drop-path sites; as
    `new_empty(B).bernoulli_(keep).div_(keep)` each is two 4-us kernels in a stream of much larger ones (56 launches per
    step).  Inside a forward epoch the sites are served rows of `floor(keep_i + U[0,1)) / keep_i` - timm's own DropPath
    formula, drawn for all sites at once - in the order the previous forward of the same module asked for them; a site
    that does not match the recorded sequence (another batch size, another model path) ends the pooling for that forward
    and draws on its own.  Off for modules that recompute activations (with_cp): the recomputation replays torch's RNG
    state, which a pooled draw made at the start of the forward is not part of."""

    def __init__(self):
        self.rows = self.seq = self.module = None
        self.pos, self.recording = 0, []

    def begin(self, module, device):
        self.module, self.recording, self.pos, self.rows = module, [], 0, None
        d = module.__dict__
        self.seq = d.get('_vah_drop_trace')
        if 'vah_drop_pool_ok' not in d:
            d['vah_drop_pool_ok'] = not any(getattr(m, 'with_cp', False) for m in module.modules())
        if not (self.seq and d['vah_drop_pool_ok'] and ENABLED.get('drop_pool', True)):
            self.seq = None
            return
        keeps = d.get('_vah_drop_keeps')
        if keeps is None or keeps.device != device or keeps.shape[0] != len(self.seq):
            keeps = d['_vah_drop_keeps'] = torch.tensor([k for k, _ in self.seq], dtype=torch.float32, device=device)[:, None]
        B = self.seq[0][1]
        if any(b != B for _, b in self.seq):
            self.seq = None
            return
        self.rows = torch.rand((len(self.seq), B), device=device).add_(keeps).floor_().div_(keeps)

    def take(self, x, keep):
        B = x.shape[0]
        if self.module is not None:
            self.recording.append((keep, B))
            if self.rows is not None:
                if self.pos < len(self.seq) and self.seq[self.pos] == (keep, B) and self.rows.device == x.device:
                    self.pos += 1
                    return self.rows[self.pos - 1]
                self.rows = None            # not the recorded sequence: every site of this forward draws on its own
        return x.new_empty((B,)).bernoulli_(keep).div_(keep)

    def end(self):
        if self.module is not None:
            d = self.module.__dict__
            if d.get('_vah_drop_trace') != self.recording:
                d['_vah_drop_trace'] = self.recording
                d.pop('_vah_drop_keeps', None)
        self.module = self.rows = self.seq = None


DROP_POOL = _DropPool()


class forward_epoch:
    """``with fused.forward_epoch(model): ...`` around a model's forward: on entry ONE multi-tensor cast
    makes the bf16 copies of all its fp32 nn.Linear parameters, which then serve every fused Linear of
    this forward; on exit (also by an exception) the epoch closes and later uses cast on their own, so
    a parameter written between two forwards - by any means, `.data` included - is always seen."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        module = self.module
        if not (ENABLED['linear'] and _bf16_autocast()):
            return self
        # the list is cached on the module; a parameter that is replaced later is simply not part of the
        # bulk cast and gets its own cast on use (slower, never stale)
        params = module.__dict__.get('_vah_linear_params')
        if params is None:
            params = [p for m in module.modules() if isinstance(m, torch.nn.Linear)
                      for p in (m.weight, m.bias) if p is not None]
            module.__dict__['_vah_linear_params'] = params
        live = [p for p in params if p.dtype == torch.float32 and p.is_cuda]
        if live:
            BF16_COPIES.begin(live)
            SIDE.begin_epoch()
            if module.training:
                DROP_POOL.begin(module, live[0].device)
        return self

    def __exit__(self, *exc):
        BF16_COPIES.end()
        DROP_POOL.end()
        return False


_GEMM_WS_BYTES = 32 << 20
WGRAD_F32 = os.environ.get('VAH_LINEAR_WGRAD', 'f32') == 'f32'
GEMM_TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tuning',
                          'gemm_table_mi355x_base_det_1024.txt')


def _configure_gemm():
    """VAH_GEMM_TUNING = mode[,candidates] (0 first heuristic answer, 1 time the heuristic
    candidates [default], 2 time every algorithm);  VAH_GEMM_TABLE = path of a tuned table to load
    ('' = none; default: the committed table);  VAH_GEMM_TABLE_DUMP = path to write the table to
    at exit."""
    spec = os.environ.get('VAH_GEMM_TUNING')
    if spec:
        parts = [int(v) for v in spec.split(',')]
        _vah.check(_vah.lib.vah_gemm_set_tuning(parts[0], parts[1] if len(parts) > 1 else 32), 'gemm_set_tuning')
    dump = os.environ.get('VAH_GEMM_TABLE_DUMP')
    if dump:
        import atexit
        atexit.register(lambda: open(dump, 'w').write(_vah.gemm_table_dump()))


_configure_gemm()
_GEMM_TABLE_LOADED = False


def _load_gemm_table():
    """On the first GEMM (needs the GPU: the table is tied to the hipBLASLt build that is loaded)."""
    global _GEMM_TABLE_LOADED
    _GEMM_TABLE_LOADED = True
    path = os.environ.get('VAH_GEMM_TABLE', GEMM_TABLE)
    if path and os.path.exists(path):
        _vah.gemm_table_load(open(path).read())


def gemm_bf16(a, b, trans_a=False, trans_b=False, out_dtype=torch.bfloat16, bias=None, out=None):
    """op(a) @ op(b) for contiguous 2-D bf16 matrices on the tuned hipBLASLt dispatcher
    (csrc/gemm.hip); fp32 accumulation, bf16 or fp32 result."""
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    assert (b.shape[1] if trans_b else b.shape[0]) == K and a.is_contiguous() and b.is_contiguous()
    if out is not None:
        assert out.shape == (M, N) and out.is_contiguous()
        d, out_dtype = out, out.dtype
    else:
        d = torch.empty((M, N), dtype=out_dtype, device=a.device)
    if M == 0 or N == 0:
        return d
    if K == 0:
        return d.zero_()
    if not _GEMM_TABLE_LOADED:
        _load_gemm_table()
    ws_bytes = _GEMM_WS_BYTES
    if bias is None and K >= 4096:       # room for the fp32 partial products of a split-K run
        ws_bytes += min(64 * M * N * 4, 160 << 20)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
    epilogue = _vah.GEMM_EPI_BIAS if bias is not None else _vah.GEMM_EPI_NONE
    with _vah.on(a.device):
        _vah.check(_vah.lib.vah_gemm_bf16(
            int(trans_a), int(trans_b), M, N, K, a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1],
            d.data_ptr(), N, int(out_dtype == torch.float32), epilogue,
            bias.data_ptr() if bias is not None else None,
            int(bias is not None and bias.dtype == torch.float32), ws.data_ptr(), ws_bytes, _stream(a)), 'gemm_bf16')
    return d


def _wgrad_bgrad(g2, x2):
    """(dW, db) = (g2^T x2 in fp32, column sums of g2) of a Linear backward: column-sum partials, the
    (split-K) GEMM and ONE launch that both reduces the GEMM's slices and sums the partials."""
    import ctypes
    R, N = g2.shape
    K = x2.shape[1]
    if not _GEMM_TABLE_LOADED:
        _load_gemm_table()
    dev = g2.device
    gw = torch.empty((N, K), dtype=torch.float32, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev)
    cws = _scratch(N, dev)
    ws_bytes = _GEMM_WS_BYTES + (min(64 * N * K * 4, 160 << 20) if R >= 4096 else 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nparts = ctypes.c_int64(0)
    with _vah.on(dev):
        st = _stream(g2)
        _vah.check(_vah.lib.vah_colsum_bf16_partials(g2.data_ptr(), R, N, cws.data_ptr(), ctypes.byref(nparts), st),
                   'colsum_partials')
        _vah.check(_vah.lib.vah_gemm_bf16_fin(1, 0, N, K, R, g2.data_ptr(), N, x2.data_ptr(), K, gw.data_ptr(), K, 1,
                                              ws.data_ptr(), ws_bytes, cws.data_ptr(), nparts.value, N, gb.data_ptr(), st),
                   'gemm_bf16_fin')
    return gw, gb


class _SideStream:
    """Weight-gradient GEMMs on a second HIP stream.  In a Linear backward the input gradient is on the critical path
    (the next layer's backward waits for it), the weight / bias gradients are not: nobody reads them before the
    optimizer.  Their GEMMs reduce over the tokens into a few output tiles (768 x 768: 36 tiles for 256 CUs, split-K
    brings that to ~150 workgroups) and leave most of the chip idle, so they run beside the main stream's kernels.
      fork:  side.wait_stream(current)                       - the operands exist
      join:  current.wait_stream(side), ONCE per backward pass, from an autograd engine callback queued by the first
             fork of the pass (the engine runs callbacks after the last node: before anything can read a .grad)
    A gradient left on the side stream is NOT handed to autograd: the engine may add it to another gradient of the
    same parameter (a second use of the weight by any operator, two forwards before one backward, retain_graph) on the
    main stream the moment the node returns, which would race with the side stream.  The node returns None for it and
    the join, once the main stream is ordered behind the side stream, stores it in (or adds it to) the parameter's
    .grad itself - what AccumulateGrad does, at the end of the pass instead of in the middle.  That is only the same thing
    when nothing observes the accumulation: see may_defer.
    The operands are kept alive until the join (their memory is freed on the main stream's pool only after the main
    stream is ordered behind the side stream), the gradients are allocated while the side stream is current.
    Off when a process group exists (DDP's bucket hooks read a gradient the moment autograd produces it, and its
    .grad tensors are views into the buckets) and under HIP-graph capture (the fork / join edges cost a replayed graph more than the overlap gains).
    Measured on base_det, eager: 27.5 -> 26.7 ms per step."""

    def __init__(self):
        self.streams = {}
        self.pending = {}          # device index -> (main stream, [tensors kept alive], [(parameter, gradient)])

    def begin_epoch(self):
        for idx in list(self.pending):      # a backward pass that died before its callback: join now, never leave a fork open
            self.join(idx)

    def note(self, weight, bias):
        """Forward: weak references to the parameters whose gradients the backward may leave on the side stream.  The
        DECISION is taken at backward time (may_defer)."""
        import weakref
        if not weight.is_leaf or (bias is not None and not bias.is_leaf) or not (BF16_COPIES.epoch & 1):
            return None
        if not weight.requires_grad or (bias is not None and not bias.requires_grad):
            return None
        return tuple(weakref.ref(p) for p in (weight, bias) if p is not None)

    @staticmethod
    def _only_accumulated(p):
        """Will this backward pass do nothing with the parameter's gradient but accumulate it into .grad?  No tensor hooks
        (they see or rewrite the gradient in mid-pass), and AccumulateGrad scheduled by the engine: under
        torch.autograd.grad() gradients are RETURNED, and backward(inputs=...) may leave the parameter out."""
        if p is None or getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None):
            return False
        with torch.enable_grad():       # the parameter's AccumulateGrad node (not kept: a node that outlives its pass pins a stream)
            acc = p.view_as(p).grad_fn.next_functions[0][0]
        try:
            return bool(torch._C._will_engine_execute_node(acc))
        except RuntimeError:            # autograd.grad(), or not inside a backward pass at all
            return False

    def may_defer(self, token, dev):
        """Backward: may this node's weight / bias gradients be produced on the side stream and stored by the join?"""
        if token is None or not (ENABLED['wgrad_overlap'] and dev.type == 'cuda'):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False               # measured: ~80 fork / join edges per step cost a replayed HIP graph 0.35 ms more than the overlap gains
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():      # DDP owns the .grad tensors (bucket views) and watches them arrive
            return False
        return all(self._only_accumulated(r()) for r in token)

    def fork(self, dev, keep):
        """-> the side stream, ordered behind everything enqueued on the current stream so far."""
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        side = self.streams.get(idx)
        if side is None:
            side = self.streams[idx] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        entry = self.pending.get(idx)
        if entry is None:
            self.pending[idx] = entry = (cur, [], [])
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self.join(idx))
        entry[1].extend(keep)
        return side

    def deliver(self, dev, token, grads):
        """The side stream's gradients of this node, for the join to accumulate (same order as the token)."""
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.pending[idx][2].extend((r(), g) for r, g in zip(token, grads))

    def join(self, idx):
        entry = self.pending.pop(idx, None)
        if entry is None:
            return
        main, keep, grads = entry
        main.wait_stream(self.streams[idx])
        with torch.no_grad(), torch.cuda.stream(main):
            for p, g in grads:
                if p is None:           # the parameter died with its module before the pass ended
                    continue
                if g.dtype != p.dtype:
                    g = g.to(p.dtype)
                if p.grad is None:
                    p.grad = g.view_as(p)
                else:
                    p.grad.add_(g.view_as(p))
        keep.clear()


SIDE = _SideStream()


class _LinearBF16(torch.autograd.Function):
    """y = x W^T + b with bf16 operands and fp32 accumulation (what autocast makes of F.linear).
    Backward: dX in bf16, dW straight into fp32 from the GEMM (no bf16 rounding, no cast kernel),
    db by the column-sum kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        x2 = x2.contiguous()
        wb = BF16_COPIES.get(weight)
        y = gemm_bf16(x2, wb, trans_b=True, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x2, wb)
        ctx.has_bias = bias is not None
        ctx.side = SIDE.note(weight, bias)
        ctx.in_shape = x.shape
        ctx.in_dtype = x.dtype
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, wb = ctx.saved_tensors
        N = wb.shape[0]
        g2 = g.reshape(-1, N)
        if g2.dtype != torch.bfloat16:
            g2 = g2.to(torch.bfloat16)
        g2 = g2.contiguous()
        gx = gw = gb = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        fin = ctx.needs_input_grad[1] and want_b and WGRAD_F32 and g2.shape[0] > 0 and ENABLED['wgrad_fin']
        deferred = fin and SIDE.may_defer(ctx.side, g2.device)
        if deferred:                    # weight / bias gradients beside the input gradient, on the side stream
            with torch.cuda.stream(SIDE.fork(g2.device, (g2, x2))):
                SIDE.deliver(g2.device, ctx.side, _wgrad_bgrad(g2, x2))
        if ctx.needs_input_grad[0]:
            gx = gemm_bf16(g2, wb).view(ctx.in_shape)
            if gx.dtype != ctx.in_dtype:
                gx = gx.to(ctx.in_dtype)
        if deferred:
            pass
        elif fin:
            gw, gb = _wgrad_bgrad(g2, x2)
        else:
            if ctx.needs_input_grad[1]:
                if WGRAD_F32:
                    gw = gemm_bf16(g2, x2, trans_a=True, out_dtype=torch.float32)
                else:
                    gw = gemm_bf16(g2, x2, trans_a=True).float()
            if want_b:
                gb = torch.empty(N, dtype=torch.float32, device=g2.device)
                ws = _scratch(N, g2.device)
                with _vah.on(g2.device):
                    _vah.check(_vah.lib.vah_colsum_bf16(g2.data_ptr(), g2.shape[0], N, gb.data_ptr(),
                                                        ws.data_ptr(), _stream(g2)), 'colsum')
        return gx, gw, gb


class _PairCopies:
    """bf16 [Wa; Wb] (rows concatenated) and fp32 [ba; bb] of two Linear layers that read the same
    input: built once per forward epoch (see _Bf16Copies), on every use outside one."""

    def __init__(self):
        self.entries = {}

    def get(self, a, b):
        key = (id(a.weight), id(b.weight))
        epoch = BF16_COPIES.epoch
        e = self.entries.get(key)
        if e is not None and e[0] == epoch and (epoch & 1) and e[1].device == a.weight.device and e[3]() is a.weight:
            return e[1], e[2]
        import weakref
        with torch.no_grad():
            w = torch.cat([a.weight.detach(), b.weight.detach()], 0).to(torch.bfloat16)
            bias = torch.cat([a.bias.detach(), b.bias.detach()], 0).float()
        if len(self.entries) > 1024:
            self.entries.clear()
        self.entries[key] = (epoch, w, bias, weakref.ref(a.weight))
        return w, bias


PAIR_COPIES = _PairCopies()


class _LinearPairBF16(torch.autograd.Function):
    """(x Wa^T + ba, x Wb^T + bb) as ONE GEMM over the concatenated weights, and one input-gradient
    GEMM, one weight-gradient GEMM and one column sum in the backward.  For MSDeformAttn's
    sampling_offsets / attention_weights (ms_deform_attn.py:108-109): two skinny GEMMs (96 and 48
    outputs per level) over the same 43008 x 768 query matrix plus the add autograd needs to sum the
    two input gradients."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, pair, f32_out=False):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        x2 = x2.contiguous()
        w, bias = pair
        y = gemm_bf16(x2, w, trans_b=True, bias=bias, out_dtype=torch.float32 if f32_out else torch.bfloat16)
        na = wa.shape[0]
        ctx.save_for_backward(x2, w)
        ctx.na, ctx.in_shape, ctx.in_dtype = na, x.shape, x.dtype
        lead = x.shape[:-1]
        return y[:, :na].contiguous().view(*lead, na), y[:, na:].contiguous().view(*lead, w.shape[0] - na)

    @staticmethod
    def backward(ctx, ga, gb):
        x2, w = ctx.saved_tensors
        na = ctx.na
        nb = w.shape[0] - na
        R = x2.shape[0]
        g = torch.empty((R, na + nb), dtype=torch.bfloat16, device=x2.device)
        g[:, :na] = ga.reshape(R, na) if ga is not None else 0
        g[:, na:] = gb.reshape(R, nb) if gb is not None else 0
        gx = gw = gbias = None
        if ctx.needs_input_grad[0]:
            gx = gemm_bf16(g, w).view(ctx.in_shape)
            if gx.dtype != ctx.in_dtype:
                gx = gx.to(ctx.in_dtype)
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        want_b = ctx.needs_input_grad[2] or ctx.needs_input_grad[4]
        if want_w and want_b and R > 0 and ENABLED['wgrad_fin']:
            gw, gbias = _wgrad_bgrad(g, x2)
        else:
            if want_w:
                gw = gemm_bf16(g, x2, trans_a=True, out_dtype=torch.float32)
            if want_b:
                gbias = torch.empty(na + nb, dtype=torch.float32, device=g.device)
                ws = _scratch(na + nb, g.device)
                with _vah.on(g.device):
                    _vah.check(_vah.lib.vah_colsum_bf16(g.data_ptr(), R, na + nb, gbias.data_ptr(), ws.data_ptr(),
                                                        _stream(g)), 'colsum')
        return (gx, gw[:na] if gw is not None else None, gbias[:na] if gbias is not None else None,
                gw[na:] if gw is not None else None, gbias[na:] if gbias is not None else None, None, None)


def linear_pair(lin_a, lin_b, x, f32_out=False):
    """``(lin_a(x), lin_b(x))`` for two nn.Linear layers on the same input.  f32_out: the outputs straight from the fp32
    accumulators (MSDeformAttn's sampling offsets: the reference keeps them in fp32, and d(out)/d(location) jumps at
    integer pixel coordinates, so 8-bit offsets cost the upstream gradients 0.3 - 0.7 of relative L2 on small maps)."""
    wa, wb = lin_a.weight, lin_b.weight
    if (ENABLED['linear'] and ENABLED['linear_pair'] and x.is_cuda and _bf16_autocast() and wa.dtype == torch.float32
            and wb.dtype == torch.float32 and lin_a.bias is not None and lin_b.bias is not None
            and x.dtype in (torch.bfloat16, torch.float32) and (wa.shape[0] + wb.shape[0]) % 8 == 0
            and wa.shape[1] % 8 == 0 and wa.shape[1] == wb.shape[1] and x.numel() > 0
            and type(lin_a) is torch.nn.Linear and type(lin_b) is torch.nn.Linear):
        return _LinearPairBF16.apply(x, wa, lin_a.bias, wb, lin_b.bias, PAIR_COPIES.get(lin_a, lin_b), f32_out)
    if f32_out and x.is_cuda and _bf16_autocast():
        # widths the paired GEMM does not take (a single deformable head: 8 + 4 outputs): two plain fp32 Linears
        with torch.autocast('cuda', enabled=False):
            xf = x.float()
            return torch.nn.functional.linear(xf, wa.float(), lin_a.bias), torch.nn.functional.linear(xf, wb.float(), lin_b.bias)
    return linear(lin_a, x), linear(lin_b, x)


class _PairCoreCopies:
    """bf16 rows of [sampling_offsets; attention_weights] reordered head by head - [offsets of head 0 | logits of head 0 |
    offsets of head 1 | ...] - with the fp32 bias in the same order, and the permutation (see msda_pair_core)."""

    def __init__(self):
        self.entries = {}

    def get(self, a, b, M):
        key = (id(a.weight), id(b.weight))
        epoch = BF16_COPIES.epoch
        e = self.entries.get(key)
        if e is not None and e[0] == epoch and (epoch & 1) and e[1].device == a.weight.device and e[4]() is a.weight:
            return e[1], e[2], e[3]
        import weakref
        na, nb = a.weight.shape[0], b.weight.shape[0]
        po, pl = na // M, nb // M
        dev = a.weight.device
        if e is not None and e[3][0].device == dev:
            perm = e[3]
        else:               # (permutation, its inverse): made once per module and device, not per call
            fw = torch.cat([torch.cat((torch.arange(m * po, (m + 1) * po), na + torch.arange(m * pl, (m + 1) * pl))) for m in range(M)])
            inv = torch.empty_like(fw)
            inv[fw] = torch.arange(fw.numel())
            perm = (fw.to(dev), inv.to(dev))
        with torch.no_grad():
            w = torch.cat([a.weight.detach(), b.weight.detach()], 0).index_select(0, perm[0]).to(torch.bfloat16)
            bias = torch.cat([a.bias.detach(), b.bias.detach()], 0).float().index_select(0, perm[0])
        if len(self.entries) > 1024:
            self.entries.clear()
        self.entries[key] = (epoch, w, bias, perm, weakref.ref(a.weight))
        return w, bias, perm


PAIR_CORE_COPIES = _PairCoreCopies()


class _MSDAPairCore(torch.autograd.Function):
    """MSDeformAttn's sampling_offsets / attention_weights Linear pair AND the fused deformable-attention core as one
    autograd node (ref ops/modules/ms_deform_attn.py:108-128): ONE GEMM over the two weight matrices with its output
    rows ordered head by head - (n, q, m) row = [L*P*2 offsets | L*P logits], fp32 straight from the accumulators - which
    the MSDA kernels read in place through a row stride.  What this removes per call: the two slice copies of the
    paired GEMM's output and the two that rebuild the gradient matrix, the bf16 rounding of the sampling offsets (the
    reference keeps them in fp32: ms_deform_attn_func.py:21; d(out)/d(location) jumps at integer pixel coordinates, so
    8-bit offsets cost the upstream gradients 0.2-0.35 of relative L2 on small problems), and one cache line per list
    entry in the backward's tile pass (a row's offsets and logits now sit side by side).  The gradients come back in
    bf16, written by the tile pass into the gradient matrix of the pair GEMM's backward in the same row order."""

    @staticmethod
    def forward(ctx, query, value, shapes, lsi, ref, wa, ba, wb, bb, copies, M, L, P):
        from ops.functions import ms_deform_attn_fused as mf
        K = query.shape[-1]
        x2 = query.reshape(-1, K)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        x2 = x2.contiguous()
        w, bias, perm = copies
        y = gemm_bf16(x2, w, trans_b=True, bias=bias, out_dtype=torch.float32)         # (N * Lq, M * 3 * L * P) fp32
        N, Lq = query.shape[0], query.shape[1]
        PS = 3 * L * P
        yv = y.view(N, Lq, M, PS)
        offsets = yv[..., :2 * L * P].unflatten(-1, (L, P, 2))
        logits = yv[..., 2 * L * P:]
        value = value.contiguous()
        refc = ref.detach().float().contiguous().view(Lq, -1, 2)
        out = mf.fused_forward(value, shapes, lsi, offsets, logits, PS, PS, refc, carrier=ref, token=BF16_COPIES.epoch)
        ctx.save_for_backward(x2, w, y, value, shapes, lsi, refc, perm[1])
        ctx.dims = (N, Lq, M, L, P, wa.shape[0])
        ctx.in_shape, ctx.in_dtype = query.shape, query.dtype
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x2, w, y, value, shapes, lsi, refc, inv = ctx.saved_tensors
        N, Lq, M, L, P, na = ctx.dims
        S, D = value.shape[1], value.shape[3]
        PS = 3 * L * P
        R = x2.shape[0]
        gout = gout.contiguous().to(value.dtype)
        g = torch.empty((R, M * PS), dtype=torch.bfloat16, device=x2.device)
        grad_value = torch.empty_like(value)
        ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
        with _vah.on(value.device):
            rc = _vah.lib.vah_msda_fused_backward_tiled(
                value.data_ptr(), 1, shapes.data_ptr(), lsi.data_ptr(), y.data_ptr(), y.data_ptr() + 2 * L * P * 4, 0, PS, PS,
                refc.data_ptr(), refc.shape[1], gout.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(), 1,
                g.data_ptr(), g.data_ptr() + 2 * L * P * 2, 1, PS, PS, ws.data_ptr(), ws_bytes, _stream(value))
        _vah.check(rc, 'vah_msda_fused_backward_tiled')
        gx = gw = gbias = None
        if ctx.needs_input_grad[0]:
            gx = gemm_bf16(g, w).view(ctx.in_shape)
            if gx.dtype != ctx.in_dtype:
                gx = gx.to(ctx.in_dtype)
        if any(ctx.needs_input_grad[5:9]):
            gwp, gbp = _wgrad_bgrad(g, x2)
            gw, gbias = gwp.index_select(0, inv), gbp.index_select(0, inv)
        return (gx, grad_value if ctx.needs_input_grad[1] else None, None, None, None,
                gw[:na] if gw is not None else None, gbias[:na] if gbias is not None else None,
                gw[na:] if gw is not None else None, gbias[na:] if gbias is not None else None, None, None, None, None)


def msda_pair_core_ok(mod, query, value, reference_points):
    """The one-node form of MSDeformAttn's offsets / weights pair + core: bf16 autocast on the GPU with bf16 values, the
    tile-pass backward and shapes the fused kernels cover (D == 32, P == 4, L in {1, 3, 4}), reference points shared by
    the batch."""
    import os
    a, b = mod.sampling_offsets, mod.attention_weights
    M, L, P = mod.n_heads, mod.n_levels, mod.n_points
    return (ENABLED['linear'] and ENABLED['linear_pair'] and ENABLED['pair_core'] and query.is_cuda and _bf16_autocast()
            and value.dtype == torch.bfloat16 and value.dim() == 4 and value.shape[-1] == 32 and P == 4 and L in (1, 3, 4)
            and (M * 3 * L * P) % 8 == 0            # rows of the pair GEMM / its column sums: 16-byte multiples
            and type(a) is torch.nn.Linear and type(b) is torch.nn.Linear and a.bias is not None and b.bias is not None
            and a.weight.dtype == torch.float32 and b.weight.dtype == torch.float32 and a.weight.shape[1] % 8 == 0
            and reference_points.shape[0] == 1 and reference_points.shape[-1] == 2 and reference_points.shape[2] in (1, L)
            and query.numel() > 0 and value.numel() > 0 and query.dtype in (torch.bfloat16, torch.float32)
            and os.environ.get('VAH_MSDA_FUSED', '1') != '0' and os.environ.get('VAH_MSDA_TILED', '1') != '0'
            and _vah.lib.vah_msda_tile_ws_bytes(value.shape[0], value.shape[1], M, L, query.shape[1], P) >= 0)


def msda_pair_core(mod, query, value, shapes, lsi, reference_points):
    a, b = mod.sampling_offsets, mod.attention_weights
    return _MSDAPairCore.apply(query, value, shapes, lsi, reference_points, a.weight, a.bias, b.weight, b.bias,
                               PAIR_CORE_COPIES.get(a, b, mod.n_heads), mod.n_heads, mod.n_levels, mod.n_points)


class _Conv1x1BF16(torch.autograd.Function):
    """1x1 convolution without bias on NCHW bf16 as plain GEMMs on the planes: out[b] (Co x HW) =
    W (Co x Ci) x[b] (Ci x HW) - no NCHW <-> NHWC conversions around an implicit-GEMM kernel; the
    weight gradient reduces over HW (65536 at the stride-4 map) with split-K."""

    @staticmethod
    def forward(ctx, x, weight):
        B, Ci, H, W = x.shape
        Co = weight.shape[0]
        x = x.contiguous()
        wb = BF16_COPIES.get(weight).view(Co, Ci)
        out = torch.empty((B, Co, H, W), dtype=torch.bfloat16, device=x.device)
        for b in range(B):
            gemm_bf16(wb, x[b].view(Ci, H * W), out=out[b].view(Co, H * W))
        ctx.save_for_backward(x, wb)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = wb.shape[0]
        g = g.contiguous().to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            for b in range(B):
                gemm_bf16(wb, g[b].view(Co, H * W), trans_a=True, out=dx[b].view(Ci, H * W))
        if ctx.needs_input_grad[1]:
            for b in range(B):
                part = gemm_bf16(g[b].view(Co, H * W), x[b].view(Ci, H * W), trans_b=True, out_dtype=torch.float32)
                dw = part if dw is None else dw.add_(part)
            dw = dw.view(Co, Ci, 1, 1)
        return dx, dw


class _UpFromTokens(torch.autograd.Function):
    """ConvTranspose2d(C, Co, kernel 2, stride 2) WITHOUT bias applied to a map given as its token rows:
    rows (B, h*w, C) -> NCHW planes (B, Co, 2h, 2w) bf16.  The transposed convolution with stride = kernel is a plain
    GEMM per image, U_b (4*Co, h*w) = Wcat (4*Co, C) rows_b^T with Wcat rows (dy, dx, co), followed by the 2 x 2 sub-pixel
    interleave (csrc/tail_ops.hip::pixel_shuffle2); the backward is the inverse interleave and two GEMMs.  MIOpen's
    NCHW transposed convolution took 494 + 846 us for this layer at base_det (2 x 768 x 128 x 128)."""

    @staticmethod
    def forward(ctx, rows, weight, h, w, addend):
        B, T, C = rows.shape
        Co = weight.shape[1]
        xb = rows.detach().to(torch.bfloat16).contiguous()
        wc = BF16_COPIES.get(weight).permute(2, 3, 1, 0).reshape(4 * Co, C).contiguous()       # rows (dy, dx, co)
        U = torch.empty((B, 4 * Co, T), dtype=torch.bfloat16, device=rows.device)
        for b in range(B):
            gemm_bf16(wc, xb[b], trans_b=True, out=U[b])
        out = torch.empty((B, Co, 2 * h, 2 * w), dtype=torch.bfloat16, device=rows.device)
        add = addend.contiguous() if addend is not None else None
        with _vah.on(rows.device):
            _vah.check(_vah.lib.vah_pixel_shuffle2_bf16(U.data_ptr(), B, Co, h, w, out.data_ptr(), 0,
                                                        add.data_ptr() if add is not None else None, _stream(rows)), 'pixel_shuffle2')
        ctx.save_for_backward(xb, wc)
        ctx.meta = (h, w, Co, rows.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        xb, wc = ctx.saved_tensors
        h, w, Co, in_dtype = ctx.meta
        B, T, C = xb.shape
        g = g.contiguous().to(torch.bfloat16)
        dU = torch.empty((B, 4 * Co, T), dtype=torch.bfloat16, device=g.device)
        with _vah.on(g.device):
            _vah.check(_vah.lib.vah_pixel_shuffle2_bf16(g.data_ptr(), B, Co, h, w, dU.data_ptr(), 1, None, _stream(g)), 'pixel_shuffle2')
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, T, C), dtype=torch.bfloat16, device=g.device)
            for b in range(B):
                gemm_bf16(dU[b], wc, trans_a=True, out=dx[b])
            if dx.dtype != in_dtype:
                dx = dx.to(in_dtype)
        if ctx.needs_input_grad[1]:
            for b in range(B):
                part = gemm_bf16(dU[b], xb[b], out_dtype=torch.float32)
                dw = part if dw is None else dw.add_(part)
            dw = dw.view(2, 2, Co, C).permute(3, 2, 0, 1).contiguous()
        return dx, dw, None, None, (g if ctx.needs_input_grad[4] else None)      # d(out)/d(addend) = identity


def up_from_tokens(up, rows, h, w, addend=None):
    """``F.conv_transpose2d(rows.transpose(1, 2).view(B, C, h, w), up.weight, None, stride=2)`` for the backbone's
    2 x 2 / stride 2 ``up`` (vit_adapter.py:46), from the token rows of the map and without the bias (the caller folds
    it into the BatchNorm tail); ``addend`` (planes-shaped bf16, e.g. c1) is summed in by the interleave pass -
    ``up(c2) + c1`` rounded once to bf16, as autocast rounds the sum of two half tensors - so the tail reads one operand
    instead of two.  None when the GEMM form does not apply."""
    wt = up.weight
    if (ENABLED['up_gemm'] and ENABLED['linear'] and rows.is_cuda and _bf16_autocast() and isinstance(up, torch.nn.ConvTranspose2d)
            and up.kernel_size == (2, 2) and up.stride == (2, 2) and up.padding == (0, 0) and up.output_padding == (0, 0)
            and up.groups == 1 and up.dilation == (1, 1) and wt.dtype == torch.float32 and rows.dim() == 3
            and rows.shape[1] == h * w and rows.shape[2] == wt.shape[0] and w % 8 == 0 and wt.shape[0] % 8 == 0
            and wt.shape[1] % 8 == 0 and rows.dtype in (torch.float32, torch.bfloat16) and rows.numel() > 0
            and (addend is None or (addend.dtype == torch.bfloat16 and tuple(addend.shape) == (rows.shape[0], wt.shape[1], 2 * h, 2 * w)))):
        return _UpFromTokens.apply(rows, wt, h, w, addend)
    return None


def patch_embed(conv, x):
    """``conv(x).flatten(2).transpose(1, 2)`` for the patch embedding (Conv2d with kernel = stride, base/vit.py:169-190)
    as ONE GEMM on bf16 patch rows - no im2col / NCHW transposes; the rows come out in token order.  Returns
    (tokens (B, N, E) bf16, H/ps, W/ps) or None when the form does not apply (input gradient wanted, other geometry)."""
    w = conv.weight
    ps = conv.kernel_size[0]
    if (ENABLED['patch_gemm'] and ENABLED['linear'] and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
            and _bf16_autocast() and x.dim() == 4 and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (ps, ps)
            and conv.stride == (ps, ps) and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1)
            and ps % 8 == 0 and x.shape[2] % ps == 0 and x.shape[3] % ps == 0 and w.dtype == torch.float32
            and w.shape[0] % 8 == 0 and x.numel() > 0):
        B, C, H, W = x.shape
        x = x.contiguous()
        rows = torch.empty((B * (H // ps) * (W // ps), C * ps * ps), dtype=torch.bfloat16, device=x.device)
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_patchify_bf16(x.data_ptr(), B, C, H, W, ps, rows.data_ptr(), _stream(x)), 'patchify')
        y = _LinearBF16.apply(rows, w.view(w.shape[0], -1), conv.bias)
        return y.view(B, (H // ps) * (W // ps), w.shape[0]), H // ps, W // ps
    return None


def conv1x1(conv, x):
    """``F.conv2d(x, conv.weight, None)`` for a 1x1 nn.Conv2d (the SPM's fc1..fc4 without their bias,
    which the callers fold into the ops that follow)."""
    w = conv.weight
    if (ENABLED['conv1x1'] and ENABLED['linear'] and x.is_cuda and x.dtype == torch.bfloat16 and _bf16_autocast()
            and x.dim() == 4 and w.dtype == torch.float32 and w.shape[2:] == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
            and (x.shape[2] * x.shape[3]) % 8 == 0 and x.numel() > 0):
        return _Conv1x1BF16.apply(x, w)
    return F.conv2d(x, w, None)


def linear(lin, x):
    """``lin(x)`` for an nn.Linear (reference: every nn.Linear of base/vit.py, adapter_modules.py
    and ms_deform_attn.py)."""
    w = lin.weight
    if (ENABLED['linear'] and x.is_cuda and _bf16_autocast() and w.dtype == torch.float32
            and x.dtype in (torch.bfloat16, torch.float32) and w.shape[0] % 8 == 0
            and w.shape[1] % 8 == 0 and x.numel() > 0 and type(lin) is torch.nn.Linear):
        return _LinearBF16.apply(x, w, lin.bias)
    return lin(x)


class _ScaleResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z, gamma, s):
        B, C = x.shape[0], x.shape[-1]
        rpb = x.numel() // (B * C)
        x, z = x.contiguous(), z.contiguous()
        y = torch.empty_like(x)
        gp = gamma.contiguous() if gamma is not None else None
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_scale_residual_fwd(
                x.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                s.data_ptr() if s is not None else None, B, rpb, C, y.data_ptr(), _stream(x)),
                'scale_residual_fwd')
        ctx.save_for_backward(z, gp, s)
        ctx.dims = (B, rpb, C)
        return y

    @staticmethod
    def backward(ctx, g):
        z, gp, s = ctx.saved_tensors
        B, rpb, C = ctx.dims
        g = g.contiguous()
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=g.device) if gp is not None else None
        ws = _scratch(C, g.device) if gp is not None else None
        with _vah.on(g.device):
            _vah.check(_vah.lib.vah_scale_residual_bwd(
                g.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                s.data_ptr() if s is not None else None, B, rpb, C, dz.data_ptr(),
                dgamma.data_ptr() if dgamma is not None else None,
                ws.data_ptr() if ws is not None else None, _stream(g)), 'scale_residual_bwd')
        return g, dz, dgamma, None


def residual(x, z, gamma=None, drop_path=None):
    """``x + drop_path(gamma * z)`` (gamma / drop_path optional), the residual update of the
    reference's Block / Injector / Extractor."""
    prob = float(getattr(drop_path, 'drop_prob', 0.) or 0.)
    training = bool(getattr(drop_path, 'training', False))
    if (ENABLED['residual'] and x.is_cuda and x.dtype == torch.float32 and z.dtype == torch.bfloat16
            and x.shape == z.shape and x.shape[-1] % 4 == 0 and x.numel() > 0
            and (gamma is None or gamma.dtype == torch.float32)):
        s = None
        if prob > 0. and training:
            s = DROP_POOL.take(x, 1.0 - prob)
        return _ScaleResidual.apply(x, z, gamma, s)
    t = gamma * z if gamma is not None else z
    return x + (drop_path(t) if drop_path is not None else t)


class _ResidualLN(torch.autograd.Function):
    """(t, h) = (x + s * gamma * z, LayerNorm(t)) in one pass; the backward sums the gradient of t
    along the residual stream and through the LayerNorm in-kernel and emits dz, dgamma with it."""

    @staticmethod
    def forward(ctx, x, z, gamma, s, weight, bias, eps):
        B, C = x.shape[0], x.shape[-1]
        rpb = x.numel() // (B * C)
        x, z = x.contiguous(), z.contiguous()
        t = torch.empty_like(x)
        h = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        rows = B * rpb
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        gp = gamma.contiguous() if gamma is not None else None
        w, b = weight.contiguous(), bias.contiguous()
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_residual_layernorm_fwd(
                x.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                s.data_ptr() if s is not None else None, B, rpb, C, w.data_ptr(), b.data_ptr(), float(eps),
                t.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream(x)), 'residual_layernorm_fwd')
        ctx.save_for_backward(t, z, gp, s, w, mean, rstd)
        ctx.dims = (B, rpb, C)
        ctx.set_materialize_grads(False)
        return t, h

    @staticmethod
    def backward(ctx, gt, gh):
        t, z, gp, s, w, mean, rstd = ctx.saved_tensors
        B, rpb, C = ctx.dims
        dev = t.device
        if gt is not None:
            gt = gt.contiguous().float()
        if gh is None:                       # the normalised copy was not used: plain residual backward
            if gt is None:
                return (None,) * 7
            dz = torch.empty_like(z)
            dgamma = torch.empty(C, dtype=torch.float32, device=dev) if gp is not None else None
            ws = _scratch(C, dev) if gp is not None else None
            with _vah.on(dev):
                _vah.check(_vah.lib.vah_scale_residual_bwd(
                    gt.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                    s.data_ptr() if s is not None else None, B, rpb, C, dz.data_ptr(),
                    dgamma.data_ptr() if dgamma is not None else None,
                    ws.data_ptr() if ws is not None else None, _stream(gt)), 'scale_residual_bwd')
            return gt, dz, dgamma, None, None, None, None
        gh = gh.contiguous().to(torch.bfloat16)
        dt = torch.empty_like(t)
        dz = torch.empty_like(z)
        grads = torch.empty(3, C, dtype=torch.float32, device=dev)
        ws = _scratch(3 * C, dev)
        with _vah.on(dev):
            _vah.check(_vah.lib.vah_residual_layernorm_bwd(
                t.data_ptr(), gh.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                gt.data_ptr() if gt is not None else None, z.data_ptr(),
                gp.data_ptr() if gp is not None else None, s.data_ptr() if s is not None else None, B, rpb, C,
                dt.data_ptr(), dz.data_ptr(), grads[2].data_ptr() if gp is not None else None,
                grads[0].data_ptr(), grads[1].data_ptr(), ws.data_ptr(), _stream(t)), 'residual_layernorm_bwd')
        return dt, dz, grads[2] if gp is not None else None, None, grads[0], grads[1], None


def _drop_path_scale(x, drop_path):
    prob = float(getattr(drop_path, 'drop_prob', 0.) or 0.)
    if prob > 0. and bool(getattr(drop_path, 'training', False)):
        return DROP_POOL.take(x, 1.0 - prob)
    return None


def residual_ln(x, z, gamma, drop_path, norm):
    """``t = x + drop_path(gamma * z); return t, norm(t)`` - a residual update followed by the
    LayerNorm of the next sub-block (base/vit.py:301-306), one pass over the rows each way."""
    if (ENABLED['residual'] and ENABLED['residual_ln'] and _ln_fusable(norm, x) and z.dtype == torch.bfloat16
            and x.shape == z.shape
            and x.dim() >= 2 and (gamma is None or gamma.dtype == torch.float32) and x.shape[-1] <= 1024):
        return _ResidualLN.apply(x, z, gamma, _drop_path_scale(x, drop_path), norm.weight, norm.bias, norm.eps)
    return layer_norm_keep(norm, residual(x, z, gamma, drop_path))


class _DWConvTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, H, W):
        B, N, C = x.shape
        x = x.contiguous()
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_bf16(
                x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, B, H, W, C, 0,
                y.data_ptr(), _stream(x)), 'dwconv_fwd')
        ctx.save_for_backward(x, w)
        ctx.dims = (B, H, W, C, bias is not None, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, H, W, C, has_bias, wdtype = ctx.dims
        g = g.contiguous().to(torch.bfloat16)
        dx = torch.empty_like(x)
        dw = torch.empty(C * 9 + C, dtype=torch.float32, device=x.device)
        ws = _scratch(10 * C, x.device)
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_bf16(
                g.data_ptr(), w.data_ptr(), None, B, H, W, C, 1, dx.data_ptr(), _stream(x)), 'dwconv_dgrad')
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_wgrad_bf16(
                x.data_ptr(), g.data_ptr(), B, H, W, C, dw.data_ptr(),
                dw[C * 9:].data_ptr() if has_bias else None, ws.data_ptr(), _stream(x)), 'dwconv_wgrad')
        return (dx, dw[:C * 9].view(C, 1, 3, 3).to(wdtype),
                dw[C * 9:].to(wdtype) if has_bias else None, None, None)


def dwconv_tokens(conv, x, H, W):
    """ConvFFN's DWConv on the concatenated token maps; None when the fused path does not apply."""
    B, N, C = x.shape
    if (ENABLED['dwconv'] and x.is_cuda and x.dtype == torch.bfloat16 and C % 4 == 0 and C <= 1024
            and H % 2 == 0 and W % 2 == 0 and N == 21 * (H // 2) * (W // 2) and x.numel() > 0
            and conv.weight.shape == (C, 1, 3, 3)):
        return _DWConvTokens.apply(x, conv.weight, conv.bias, H, W)
    return None


# ---------------------------------------------------------------------------------------
# output tail: BatchNorm(a + b + bilinear_upsample(x))
# ---------------------------------------------------------------------------------------
def _sync_group(norm):
    """Process group whose ranks share the statistics of a SyncBatchNorm in training (None: local)."""
    import torch.distributed as dist
    if not isinstance(norm, torch.nn.SyncBatchNorm) or not dist.is_available() or not dist.is_initialized():
        return None
    group = norm.process_group if norm.process_group is not None else dist.group.WORLD
    if dist.get_world_size(group) > 1:
        return group
    from . import data_parallel
    return group if data_parallel.one_rank_group() else None        # one rank: the collectives run only on request


class _BNTail(torch.autograd.Function):
    """y = [relu] BatchNorm(a + b + upsample(x)): statistics pass, one bookkeeping launch (mean, rstd,
    running statistics), normalise pass; SyncBatchNorm all-reduces the sums in between."""

    @staticmethod
    def forward(ctx, a, b, x, weight, bias, shift, norm, scale, relu, out_dtype):
        N, C, H, W = a.shape
        sh = shift.detach().float().contiguous() if shift is not None else None
        shp = sh.data_ptr() if sh is not None else None
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        x = x.contiguous().float() if x is not None else None
        dev, st = a.device, _stream(a)
        ops = (a.data_ptr(), int(a.dtype == torch.bfloat16), b.data_ptr() if b is not None else None,
               int(b is not None and b.dtype == torch.bfloat16), x.data_ptr() if x is not None else None,
               scale, N, C, H, W)
        training = norm.training or norm.running_mean is None
        group = _sync_group(norm) if training else None
        w = weight.detach().float().contiguous() if weight is not None else None
        bb = bias.detach().float().contiguous() if bias is not None else None
        with _vah.on(dev):
            if training:
                sums = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
                ws = torch.empty(_vah.lib.vah_bn_tail_ws_floats(C), dtype=torch.float32, device=dev)
                _vah.check(_vah.lib.vah_bn_tail_stats(*ops, shp, sums.data_ptr(), ws.data_ptr(), st), 'bn_tail_stats')
                sums[2 * C:].fill_(float(N * H * W))
                if group is not None:
                    import torch.distributed as dist
                    dist.all_reduce(sums, group=group)
                mean = torch.empty(C, dtype=torch.float32, device=dev)
                rstd = torch.empty(C, dtype=torch.float32, device=dev)
                track = norm.running_mean is not None
                _vah.check(_vah.lib.vah_bn_finalize_stats(
                    sums.data_ptr(), C, float(norm.eps), float(norm.momentum),
                    norm.running_mean.data_ptr() if track else None, norm.running_var.data_ptr() if track else None,
                    mean.data_ptr(), rstd.data_ptr(), st), 'bn_finalize_stats')
                if track:
                    with torch.no_grad():
                        norm.num_batches_tracked += 1
                count = sums[2 * C:]
            else:
                count = None
                mean = norm.running_mean.float().contiguous()
                rstd = torch.rsqrt(norm.running_var.float() + norm.eps)
            y = torch.empty((N, C, H, W), dtype=out_dtype, device=dev)
            _vah.check(_vah.lib.vah_bn_tail_apply(
                *ops, mean.data_ptr(), rstd.data_ptr(), w.data_ptr() if w is not None else None,
                bb.data_ptr() if bb is not None else None, int(relu), shp, y.data_ptr(),
                int(out_dtype == torch.bfloat16), st), 'bn_tail_apply')
        ctx.save_for_backward(a, b, x, mean, rstd, w, bb, count, sh)
        ctx.meta = (scale, training, group, weight is not None, bias is not None, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, x, mean, rstd, w, bb, count, sh = ctx.saved_tensors
        scale, training, group, has_w, has_b, relu = ctx.meta
        shp = sh.data_ptr() if sh is not None else None
        N, C, H, W = a.shape
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dy_bf16 = int(dy.dtype == torch.bfloat16)
        dev, st = a.device, _stream(a)
        ops = (a.data_ptr(), int(a.dtype == torch.bfloat16), b.data_ptr() if b is not None else None,
               int(b is not None and b.dtype == torch.bfloat16), x.data_ptr() if x is not None else None,
               scale, N, C, H, W)
        wp = w.data_ptr() if w is not None else None
        bp = bb.data_ptr() if bb is not None else None
        with _vah.on(dev):
            sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
            ws = torch.empty(_vah.lib.vah_bn_tail_ws_floats(C), dtype=torch.float32, device=dev)
            _vah.check(_vah.lib.vah_bn_tail_bwd_stats(*ops, mean.data_ptr(), rstd.data_ptr(), wp, bp, int(relu), shp,
                                                      dy.data_ptr(), dy_bf16, sums.data_ptr(), ws.data_ptr(), st),
                       'bn_tail_bwd_stats')
            local = sums.clone() if (training and group is not None) else sums      # dweight / dbias are per-rank sums
            dweight = local[C:] if has_w else None
            dbias = local[:C] if has_b else None
            if training:
                if group is not None:
                    import torch.distributed as dist
                    dist.all_reduce(sums, group=group)
                means = sums / count
            else:
                means = torch.zeros_like(sums)          # running statistics are constants
            need_a, need_b, need_x = ctx.needs_input_grad[:3]
            da = torch.empty_like(a) if need_a else None
            db = torch.empty_like(b) if (b is not None and need_b) else None
            dx = None
            if x is not None and need_x:
                dx = torch.zeros_like(x) if scale > 1 else torch.empty_like(x)
            if da is not None or db is not None or dx is not None:
                _vah.check(_vah.lib.vah_bn_tail_bwd_apply(
                    *ops, mean.data_ptr(), rstd.data_ptr(), wp, bp, int(relu), shp, dy.data_ptr(), dy_bf16,
                    means[:C].data_ptr(), means[C:].data_ptr(),
                    da.data_ptr() if da is not None else None, db.data_ptr() if db is not None else None,
                    dx.data_ptr() if dx is not None else None, st), 'bn_tail_bwd_apply')
        dshift = None
        if sh is not None and ctx.needs_input_grad[5]:
            # d/d(shift) = sum of dt over the channel: BatchNorm in training removes channel constants
            # (exactly 0); with running statistics it is gamma * rstd * sum(dy)
            dshift = torch.zeros_like(sh) if training else (w if w is not None else 1.0) * rstd * local[:C]
        return da, db, dx, dweight, dbias, dshift, None, None, None, None


def _bn_fusable(norm, a):
    return (isinstance(norm, torch.nn.modules.batchnorm._BatchNorm) and a.is_cuda and _bf16_autocast()
            and a.dim() == 4 and a.dtype in (torch.bfloat16, torch.float32) and a.shape[3] <= 8192
            and a.numel() > 0 and norm.momentum is not None
            and (not norm.training or a.shape[0] * a.shape[2] * a.shape[3] > 1)
            and (norm.training or norm.running_mean is not None)
            and (norm.weight is None or norm.weight.dtype == torch.float32))


def tail_takes_conv_bias(norm, ref):
    """True when bn_tail will run fused for inputs like ``ref``: the caller may then run the
    convolutions that feed it WITHOUT bias and hand the biases to bn_tail as ``shift``."""
    return ENABLED['bn_tail'] and ENABLED['bias_fold'] and _bn_fusable(norm, ref)


def halve(x):
    """``F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=False)`` (vit_adapter.py:121).
    With even H, W the taps are (0.5, 0.5) in both directions: the 2x2 mean.  torch's bilinear
    kernel parallelises over output pixels only (706 us for 2x768x32x32 on MI355X); on the bf16
    GPU path the mean is taken with avg_pool2d, elsewhere the reference call is kept."""
    if (ENABLED['bn_tail'] and x.is_cuda and _bf16_autocast() and x.dim() == 4 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0):
        return F.avg_pool2d(x.float(), 2)
    return F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=False)


def bn_tail(norm, a, b=None, x=None, scale=1, shift=None):
    """``norm(a + b + F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False))``
    for a (Sync)BatchNorm2d ``norm`` - the output tail of the backbone (ref vit_adapter.py:106-127);
    ``b`` / ``x`` optional, ``scale == 1`` adds ``x`` as it is.  ``shift`` (C,): per-channel constant
    added to the sum (biases of the convolutions that made ``a`` / ``b``, applied here for free)."""
    if (ENABLED['bn_tail'] and _bn_fusable(norm, a) and x is not None
            and (b is None or (b.shape == a.shape and b.dtype in (torch.bfloat16, torch.float32)))
            and scale in (1, 2, 4, 8) and a.shape[3] % (4 * scale) == 0 and a.shape[2] % scale == 0
            and tuple(x.shape) == (a.shape[0], a.shape[1], a.shape[2] // scale, a.shape[3] // scale)
            and x.dtype in (torch.bfloat16, torch.float32)):
        return _BNTail.apply(a, b, x, norm.weight, norm.bias, shift, norm, scale, False, torch.float32)
    t = a if b is None else a + b
    if shift is not None:
        t = t + shift.view(1, -1, 1, 1).to(t.dtype)
    if x is not None:
        t = t + (x if scale == 1 else F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False))
    return norm(t)


BN_RELU_MIN_NUMEL = int(os.environ.get('VAH_BN_RELU_MIN_NUMEL', 8 << 20))


def bn_relu(norm, a):
    """``relu(norm(a))`` for the conv -> SyncBatchNorm -> ReLU triples of the SpatialPriorModule
    (adapter_modules.py:217-241): statistics pass + normalise-and-clamp pass, output in a's dtype;
    the backward recomputes the ReLU mask from ``a``."""
    # the two-pass form pays from a few million elements on (below that MIOpen's single kernel wins)
    if ENABLED['bn_relu'] and _bn_fusable(norm, a) and a.shape[3] % 4 == 0 and a.numel() >= BN_RELU_MIN_NUMEL:
        return _BNTail.apply(a, None, None, norm.weight, norm.bias, None, norm, 1, True, a.dtype)
    return F.relu(norm(a))


# ---------------------------------------------------------------------------------------
# pyramid assembly: token sequences -> NCHW maps
# ---------------------------------------------------------------------------------------
class _TokensToMaps(torch.autograd.Function):
    """(B, T, C) fp32 tokens holding consecutive maps of sizes ``hw`` -> one (B, C, h, w) map each.
    Backward writes every map's gradient straight into its token range of ONE gradient tensor
    (autograd's route: per map a transposed copy, a zero-filled full-size gradient, a slice copy and
    an add)."""

    @staticmethod
    def forward(ctx, tokens, hw):
        B, T, C = tokens.shape
        tokens = tokens.contiguous()
        outs, t0 = [], 0
        with _vah.on(tokens.device):
            for h, w in hw:
                o = torch.empty((B, C, h, w), dtype=torch.float32, device=tokens.device)
                _vah.check(_vah.lib.vah_transpose_tokens(tokens.data_ptr(), B, T, t0, h * w, C, o.data_ptr(), 1, 0, None,
                                                         _stream(tokens)), 'transpose_tokens')
                outs.append(o)
                t0 += h * w
        ctx.hw, ctx.shape = hw, (B, T, C)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        B, T, C = ctx.shape
        dev = next(g.device for g in grads if g is not None)
        gt = torch.empty((B, T, C), dtype=torch.float32, device=dev)
        t0 = 0
        with _vah.on(dev):
            for (h, w), g in zip(ctx.hw, grads):
                if g is None:
                    gt[:, t0:t0 + h * w].zero_()
                else:
                    g = g.contiguous().float()
                    _vah.check(_vah.lib.vah_transpose_tokens(g.data_ptr(), B, T, t0, h * w, C, gt.data_ptr(), 0, 0, None,
                                                             _stream(g)), 'transpose_tokens')
                t0 += h * w
        return gt, None


def tokens_to_maps(tokens, hw):
    """``[tokens[:, a:b].transpose(1, 2).reshape(B, C, h, w).contiguous() for the consecutive ranges]``
    (vit_adapter.py:113-119); ``hw``: list of (h, w) whose areas add up to the token count."""
    hw = tuple((int(h), int(w)) for h, w in hw)
    assert sum(h * w for h, w in hw) == tokens.shape[1]
    if (ENABLED['maps'] and tokens.is_cuda and tokens.dtype == torch.float32 and tokens.dim() == 3
            and tokens.numel() > 0 and tokens.shape[0] <= 65535):
        return list(_TokensToMaps.apply(tokens, hw))
    outs, t0 = [], 0
    B, _, C = tokens.shape
    for h, w in hw:
        outs.append(tokens[:, t0:t0 + h * w].transpose(1, 2).reshape(B, C, h, w).contiguous())
        t0 += h * w
    return outs


class _MapsToTokens(torch.autograd.Function):
    """cat_l(map_l.flatten(2).transpose(1, 2) + vec_l) -> (B, sum T_l, C) fp32 in one pass per map; the
    backward hands each map its gradient already transposed (bf16 like the map) and the vectors the
    column sums of their token ranges."""

    @staticmethod
    def forward(ctx, *args):
        n = len(args) // 2
        maps, vecs = args[:n], args[n:]
        B, C = maps[0].shape[:2]
        hw = [(m.shape[2], m.shape[3]) for m in maps]
        T = sum(h * w for h, w in hw)
        dev = maps[0].device
        out = torch.empty((B, T, C), dtype=torch.float32, device=dev)
        t0 = 0
        with _vah.on(dev):
            for m, v, (h, w) in zip(maps, vecs, hw):
                m = m.contiguous()
                vv = v.detach().float().contiguous() if v is not None else None
                _vah.check(_vah.lib.vah_transpose_tokens(
                    m.data_ptr(), B, T, t0, h * w, C, out.data_ptr(), 0, int(m.dtype == torch.bfloat16),
                    vv.data_ptr() if vv is not None else None, _stream(m)), 'transpose_tokens')
                t0 += h * w
        ctx.meta = (hw, [m.dtype for m in maps], [v is not None for v in vecs], (B, T, C))
        return out

    @staticmethod
    def backward(ctx, g):
        hw, dts, has_vec, (B, T, C) = ctx.meta
        g = g.contiguous().float()
        gmaps, gvecs, t0 = [], [], 0
        with _vah.on(g.device):
            for (h, w), dt, hv in zip(hw, dts, has_vec):
                gm = torch.empty((B, C, h, w), dtype=dt, device=g.device)
                _vah.check(_vah.lib.vah_transpose_tokens(g.data_ptr(), B, T, t0, h * w, C, gm.data_ptr(), 1,
                                                         int(dt == torch.bfloat16), None, _stream(g)), 'transpose_tokens')
                gmaps.append(gm)
                gv = None
                if hv and C % 4 == 0:
                    gv = torch.empty(C, dtype=torch.float32, device=g.device)
                    ws = _scratch(C, g.device)
                    _vah.check(_vah.lib.vah_colsum_f32(g[:, t0:].data_ptr(), B, T * C, h * w, C, gv.data_ptr(),
                                                       ws.data_ptr(), _stream(g)), 'colsum_f32')
                elif hv:
                    gv = g[:, t0:t0 + h * w].sum((0, 1))
                gvecs.append(gv)
                t0 += h * w
        return (*gmaps, *gvecs)


def maps_to_tokens(maps, vecs):
    """``torch.cat([m.flatten(2).transpose(1, 2) + v for m, v in zip(maps, vecs)], dim=1)`` in fp32
    (the SPM's c2..c4 maps with their conv bias + level embedding, vit_adapter.py:94-97)."""
    if (ENABLED['maps'] and maps[0].is_cuda and all(m.dim() == 4 and m.dtype in (torch.bfloat16, torch.float32)
                                                    and m.shape[:2] == maps[0].shape[:2] for m in maps)
            and maps[0].numel() > 0 and maps[0].shape[0] <= 65535):
        return _MapsToTokens.apply(*maps, *vecs)
    return torch.cat([m.flatten(2).transpose(1, 2).float() + (v if v is not None else 0.) for m, v in zip(maps, vecs)], dim=1)


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        x = x.contiguous()
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=torch.bfloat16, device=x.device)
        idx = torch.empty((N, C, Ho, Wo), dtype=torch.uint8, device=x.device)
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_maxpool3s2_fwd_bf16(x.data_ptr(), N * C, H, W, y.data_ptr(), idx.data_ptr(),
                                                        _stream(x)), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous().to(torch.bfloat16)
        gx = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=gy.device)
        with _vah.on(gy.device):
            _vah.check(_vah.lib.vah_maxpool3s2_bwd_bf16(gy.data_ptr(), idx.data_ptr(), N * C, H, W, gx.data_ptr(),
                                                        _stream(gy)), 'maxpool_bwd')
        return gx


def max_pool(pool, x):
    """``pool(x)`` for the SPM stem's nn.MaxPool2d(kernel_size=3, stride=2, padding=1)."""
    def _is(v, k):
        return v == k or v == (k, k)
    if (ENABLED['maxpool'] and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.numel() > 0
            and isinstance(pool, torch.nn.MaxPool2d) and _is(pool.kernel_size, 3) and _is(pool.stride, 2)
            and _is(pool.padding, 1) and _is(pool.dilation, 1) and not pool.ceil_mode and not pool.return_indices):
        return _MaxPool3s2.apply(x)
    return pool(x)

