"""ops.vae — pooling / upsampling, Gaussian heads, 2x2 phase decomposition of strided layers, weight splits, concat_elu, resampling.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import FUSE_SKIP, _chk, _dense_per_image, _p, _sink, _stream, zeros, zeros_like


# --------------------------------------------------------------------------------------------
# VAE pieces
# --------------------------------------------------------------------------------------------
class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_skip):
        lib = _lib.load()
        x = _chk(x, "avgpool2.x")
        n, c, h, w = x.shape
        if h % 2 or w % 2:
            raise ValueError("avg_pool2: H and W must be even")
        y = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
        _lib.check(lib.pg_avgpool2_fwd(x.data_ptr(), y.data_ptr(), n * c, h // 2, w // 2, _stream()),
                   "pg_avgpool2_fwd")
        if n_skip:
            # one pass-through alias of x for its other readers: their gradient comes back to THIS node and is added by the
            # pooling's own backward kernel (the protocol of _ConvTaps' n_skip)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, d_skip=None):
        lib = _lib.load()
        dy = _chk(dy, "avgpool2.dy")
        n, c, oh, ow = dy.shape
        dx = torch.empty((n, c, 2 * oh, 2 * ow), device=dy.device, dtype=torch.float32)
        if d_skip is not None:
            d_skip = _chk(d_skip, "avgpool2.d_skip")
            _lib.check(lib.pg_avgpool2_bwd_res(dy.data_ptr(), d_skip.data_ptr(), dx.data_ptr(), n * c, oh, ow, _stream()),
                       "pg_avgpool2_bwd_res")
        else:
            _lib.check(lib.pg_avgpool2_bwd(dy.data_ptr(), dx.data_ptr(), n * c, oh, ow, _stream()),
                       "pg_avgpool2_bwd")
        return dx, None


def avg_pool2(x, n_skip=0):
    """nn.AvgPool2d(kernel_size=2, stride=2). n_skip=1 (extension) returns (y, x_alias): x_alias is x for the caller's other
    readers, whose gradient the pooling's backward kernel adds (no gradient-sum kernel of autograd)."""
    if n_skip not in (0, 1):
        raise ValueError("avg_pool2: n_skip is 0 or 1")
    if n_skip and not (FUSE_SKIP and x.requires_grad):
        return _AvgPool2.apply(x, 0), x
    return _AvgPool2.apply(x, int(n_skip))


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _chk(x, "upsample2.x")
        n, c, h, w = x.shape
        y = torch.empty((n, c, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
        _lib.check(lib.pg_upsample2_fwd(x.data_ptr(), y.data_ptr(), n * c, h, w, _stream()),
                   "pg_upsample2_fwd")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _chk(dy, "upsample2.dy")
        n, c, h2, w2 = dy.shape
        dx = torch.empty((n, c, h2 // 2, w2 // 2), device=dy.device, dtype=torch.float32)
        _lib.check(lib.pg_upsample2_bwd(dy.data_ptr(), dx.data_ptr(), n * c, h2 // 2, w2 // 2, _stream()),
                   "pg_upsample2_bwd")
        return dx


def upsample2_nearest(x):
    """nn.Upsample(scale_factor=2, mode="nearest")"""
    return _Upsample2.apply(x)


class _GaussHead(torch.autograd.Function):
    """mode 0: (q, eps) -> z, kl vs N(0,1); mode 1: (q, p, eps) -> z, kl(q||p); mode 2: (p, eps) -> z."""

    @staticmethod
    def forward(ctx, q, p, eps, latent, mode, split_rest=False):
        """split_rest (mode 1, p wider than 2 * latent channels): third output = p[:, 2 * latent:] (a view); its
        gradient is written into dp's channel range by ONE copy instead of autograd's slice backward (zero fill +
        strided copy) and a full-size add with this function's own dp."""
        lib = _lib.load()
        ref = q if q is not None else p
        n, _, h, w = ref.shape
        L = h * w
        if q is not None:
            q = _chk(q, "gauss.q")
            if q.shape[1] < 2 * latent:
                raise ValueError("gauss head: q needs 2*latent channels")
        if p is not None:
            p = _chk(p, "gauss.p")
            if p.shape[1] < 2 * latent:
                raise ValueError("gauss head: p needs 2*latent channels")
        eps = _chk(eps, "gauss.eps")
        if tuple(eps.shape) != (n, latent, h, w):
            raise ValueError(f"gauss head: eps shape {tuple(eps.shape)} != {(n, latent, h, w)}")
        z = torch.empty((n, latent, h, w), device=ref.device, dtype=torch.float32)
        kl = zeros((n,), ref.device)
        _lib.check(
            lib.pg_gauss_head_fwd(_p(q), _p(p), eps.data_ptr(), z.data_ptr(), kl.data_ptr(), n, latent,
                                  L, 0 if q is None else q.shape[1] * L,
                                  0 if p is None else p.shape[1] * L, mode, _stream()),
            "pg_gauss_head_fwd",
        )
        ctx.save_for_backward(*[t for t in (q, p, eps) if t is not None])
        ctx.cfg = (q is not None, p is not None, latent, mode)
        ctx.mark_non_differentiable(kl) if mode == 2 else None
        ctx.split_rest = bool(split_rest)
        if split_rest:
            if p is None or p.shape[1] <= 2 * latent:
                raise ValueError("gauss head: split_rest needs p with more than 2 * latent channels")
            return z, kl, p[:, 2 * latent:]
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl, d_rest=None):
        lib = _lib.load()
        has_q, has_p, latent, mode = ctx.cfg
        saved = list(ctx.saved_tensors)
        q = saved.pop(0) if has_q else None
        p = saved.pop(0) if has_p else None
        eps = saved.pop(0)
        ref = q if q is not None else p
        n, _, h, w = ref.shape
        L = h * w
        dz = _chk(dz, "gauss.dz") if dz is not None else None
        dkl = _chk(dkl, "gauss.dkl") if (dkl is not None and mode != 2) else None
        dq = dp = None
        if q is not None:
            dq = torch.empty_like(q) if q.shape[1] == 2 * latent else zeros_like(q)
        rest_direct = ctx.split_rest and d_rest is not None
        if p is not None:
            dp = torch.empty_like(p) if (p.shape[1] == 2 * latent or rest_direct) else zeros_like(p)
        _lib.check(
            lib.pg_gauss_head_bwd(_p(q), _p(p), eps.data_ptr(), _p(dz), _p(dkl), _p(dq), _p(dp), n,
                                  latent, L, 0 if q is None else q.shape[1] * L,
                                  0 if p is None else p.shape[1] * L, mode, _stream()),
            "pg_gauss_head_bwd",
        )
        if rest_direct:  # dp[:, 2 * latent:] = d_rest: rows of (C - 2 latent) * L floats, batch-strided on either side
            d_rest = d_rest if _dense_per_image(d_rest) else _chk(d_rest, "gauss.d_rest")
            rest = p.shape[1] - 2 * latent
            dst = dp[:, 2 * latent:]
            _lib.check(lib.pg_copy_rows(d_rest.data_ptr(), dst.data_ptr(), n, rest * L, d_rest.stride(0), dst.stride(0),
                                        0, _stream()), "pg_copy_rows")
        return dq, dp, None, None, None, None


def gaussian_head_unit(h, eps, latent_channels):
    """h = [mean | log_std]: returns (z = mean + exp(log_std) * eps, KL(q || N(0, 1)) summed per sample)."""
    return _GaussHead.apply(h, None, eps, latent_channels, 0)


def gaussian_head_pair(q, p, eps, latent_channels, split_rest=False):
    """Returns (z ~ q, KL(q || p) per sample); q, p hold [mean | log_std | ...] along channels.
    split_rest=True also returns p[:, 2 * latent_channels:] (see _GaussHead.forward)."""
    return _GaussHead.apply(q, p, eps, latent_channels, 1, split_rest)


def gaussian_head_prior(p, eps, latent_channels):
    """z = mean_p + exp(log_std_p) * eps."""
    return _GaussHead.apply(None, p, eps, latent_channels, 2)[0]


class _PhaseSplit(torch.autograd.Function):
    """x (N, C, 2H, 2W) -> (4, N, C, H, W) with out[2*pr+pc, n, c, r, q] = x[n, c, 2r+pr, 2q+pc]."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _chk(x, "phase_split.x")
        n, c, h2, w2 = x.shape
        if h2 % 2 or w2 % 2:
            raise ValueError("phase_split: H and W must be even")
        xs = torch.empty((4, n, c, h2 // 2, w2 // 2), device=x.device, dtype=torch.float32)
        _lib.check(lib.pg_phase_split2(x.data_ptr(), xs.data_ptr(), n * c, h2 // 2, w2 // 2, 0, _stream()),
                   "pg_phase_split2")
        return xs

    @staticmethod
    def backward(ctx, dxs):
        lib = _lib.load()
        dxs = _chk(dxs, "phase_split.dxs")
        _, n, c, h, w = dxs.shape
        dx = torch.empty((n, c, 2 * h, 2 * w), device=dxs.device, dtype=torch.float32)
        _lib.check(lib.pg_phase_split2(dx.data_ptr(), dxs.data_ptr(), n * c, h, w, 1, _stream()),
                   "pg_phase_split2")
        return dx


class _PhaseMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs):
        lib = _lib.load()
        xs = _chk(xs, "phase_merge.xs")
        _, n, c, h, w = xs.shape
        x = torch.empty((n, c, 2 * h, 2 * w), device=xs.device, dtype=torch.float32)
        _lib.check(lib.pg_phase_split2(x.data_ptr(), xs.data_ptr(), n * c, h, w, 1, _stream()),
                   "pg_phase_split2")
        return x

    @staticmethod
    def backward(ctx, dx):
        lib = _lib.load()
        dx = _chk(dx, "phase_merge.dx")
        n, c, h2, w2 = dx.shape
        dxs = torch.empty((4, n, c, h2 // 2, w2 // 2), device=dx.device, dtype=torch.float32)
        _lib.check(lib.pg_phase_split2(dx.data_ptr(), dxs.data_ptr(), n * c, h2 // 2, w2 // 2, 0, _stream()),
                   "pg_phase_split2")
        return dxs


class _PhaseWeights(torch.autograd.Function):
    """The four 2x2 phase kernels of a 4x4 / stride-2 weight: ONE launch each way (pg_phase_weights / pg_phase_weights_bwd).

    transposed=False (Conv2d, weight (Co, Ci, 4, 4)): out[2 pr + pc] = w[:, :, (1 - pr)::2, (1 - pc)::2].
    transposed=True (ConvTranspose2d, weight (Ci, Co, 4, 4)): out[2 pr + pc] =
    w.transpose(0, 1)[:, :, (1 - pr)::2, (1 - pc)::2].flip(2, 3).
    (Written as slices these were four strided copies forward and, per slice, a zero fill, a strided copy and an add into the
    weight's gradient backward; round 5 made them one permuted copy each way out of ATen's permute / flip / stack / contiguous;
    round 6: the library's own kernels, and the backward adds straight into the parameter's flat-gradient slice when it has one.)"""

    @staticmethod
    def forward(ctx, w, transposed, sink):
        if w.dim() != 4 or tuple(w.shape[2:]) != (4, 4):
            raise ValueError("phase_weights: expected a (*, *, 4, 4) weight")
        w = _chk(w, "phase_weights.w")
        ctx.transposed, ctx.shape, ctx.sink = bool(transposed), tuple(w.shape), sink
        a, b = w.shape[:2]
        co, ci = (b, a) if transposed else (a, b)
        p = torch.empty((4, co, ci, 2, 2), device=w.device, dtype=torch.float32)
        _lib.check(_lib.load().pg_phase_weights(w.data_ptr(), p.data_ptr(), a, b, int(ctx.transposed), _stream()),
                   "pg_phase_weights")
        return tuple(p[k] for k in range(4))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes

        a, b = ctx.shape[:2]
        gs = [None if g is None else _chk(g, "phase_weights.g") for g in grads]
        like = next(g for g in gs if g is not None)
        ptrs = (ctypes.c_void_p * 4)(*[None if g is None else g.data_ptr() for g in gs])
        sink = ctx.sink
        if sink is not None:  # the parameter's slice of the flat gradient buffer: add in place, no gradient tensor for autograd
            _lib.check(_lib.load().pg_phase_weights_bwd(ptrs, sink.data_ptr(), a, b, int(ctx.transposed), 1, _stream()),
                       "pg_phase_weights_bwd")
            return None, None, None
        dw = torch.empty(ctx.shape, device=like.device, dtype=torch.float32)
        _lib.check(_lib.load().pg_phase_weights_bwd(ptrs, dw.data_ptr(), a, b, int(ctx.transposed), 0, _stream()),
                   "pg_phase_weights_bwd")
        return dw, None, None


class _SplitInChannels(torch.autograd.Function):
    """(w[:, :c1], w[:, c1:]) as two contiguous tensors: two row copies (pg_copy_rows) each way; the backward adds straight
    into the parameter's flat-gradient slice when it has one (as slices: two zero fills, two strided copies and an add into
    the weight's gradient; round 5: two ATen copies + a concatenation + autograd's accumulation)."""

    @staticmethod
    def forward(ctx, w, c1, sink):
        if not 0 < c1 < w.shape[1]:
            raise ValueError("split_in_channels: split point outside the weight's input channels")
        lib = _lib.load()
        w = _chk(w, "split_in_channels.w")
        co, ci = w.shape[0], w.shape[1]
        k = int(w[0, 0].numel())  # kh * kw
        ctx.shape, ctx.c1, ctx.sink, ctx.k = tuple(w.shape), c1, sink, k
        wa = torch.empty((co, c1) + tuple(w.shape[2:]), device=w.device, dtype=torch.float32)
        wb = torch.empty((co, ci - c1) + tuple(w.shape[2:]), device=w.device, dtype=torch.float32)
        _lib.check(lib.pg_copy_rows(w.data_ptr(), wa.data_ptr(), co, c1 * k, ci * k, c1 * k, 0, _stream()), "pg_copy_rows")
        _lib.check(lib.pg_copy_rows(w.data_ptr() + 4 * c1 * k, wb.data_ptr(), co, (ci - c1) * k, ci * k, (ci - c1) * k, 0,
                                    _stream()), "pg_copy_rows")
        return wa, wb

    @staticmethod
    def backward(ctx, ga, gb):
        lib = _lib.load()
        co, ci = ctx.shape[:2]
        c1, k, sink = ctx.c1, ctx.k, ctx.sink
        like = ga if ga is not None else gb
        acc = 1 if sink is not None else 0
        dst = sink if sink is not None else zeros(ctx.shape, like.device)
        for g, off, n in ((ga, 0, c1), (gb, c1, ci - c1)):
            if g is None:
                continue
            g = _chk(g, "split_in_channels.g")
            _lib.check(lib.pg_copy_rows(g.data_ptr(), dst.data_ptr() + 4 * off * k, co, n * k, n * k, ci * k, acc, _stream()),
                       "pg_copy_rows")
        return (None if sink is not None else dst), None, None


def split_in_channels(w, c1):
    return _SplitInChannels.apply(w, int(c1), _sink(w))


def phase_weights(w, transposed=False):
    return _PhaseWeights.apply(w, transposed, _sink(w))


def phase_split(x):
    return _PhaseSplit.apply(x)


def phase_merge(xs):
    return _PhaseMerge.apply(xs)


class _PhaseMerge4(torch.autograd.Function):
    """x[n, c, 2r + pr, 2q + pc] = p[2 pr + pc][n, c, r, q] from FOUR separate tensors (no stacked copy); backward scatters the
    gradient into four tensors."""

    @staticmethod
    def forward(ctx, p0, p1, p2, p3):
        import ctypes

        ps = [_chk(t, "phase_merge4.p") for t in (p0, p1, p2, p3)]
        if any(t.shape != ps[0].shape for t in ps):
            raise ValueError("phase_merge4: shape mismatch")
        n, c, h, w = ps[0].shape
        x = torch.empty((n, c, 2 * h, 2 * w), device=ps[0].device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ps])
        _lib.check(_lib.load().pg_phase_merge4(x.data_ptr(), ptrs, n * c, h, w, 1, _stream()), "pg_phase_merge4")
        return x

    @staticmethod
    def backward(ctx, dx):
        import ctypes

        dx = _chk(dx, "phase_merge4.dx")
        n, c, h2, w2 = dx.shape
        gs = [torch.empty((n, c, h2 // 2, w2 // 2), device=dx.device, dtype=torch.float32) for _ in range(4)]
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gs])
        _lib.check(_lib.load().pg_phase_merge4(dx.data_ptr(), ptrs, n * c, h2 // 2, w2 // 2, 0, _stream()), "pg_phase_merge4")
        return tuple(gs)


class _PhaseSplit4(torch.autograd.Function):
    """x (N, C, 2H, 2W) -> FOUR tensors p[2 pr + pc][n, c, r, q] = x[n, c, 2r + pr, 2q + pc]; backward interleaves the four
    gradients in one launch. (Indexing a stacked (4, N, C, H, W) tensor instead costs autograd, per phase, a zero fill of the whole
    stack, a strided copy and an add: 2.3 % + 3 % of beta-VAE's kernel time in round 5's table.)"""

    @staticmethod
    def forward(ctx, x):
        import ctypes

        x = _chk(x, "phase_split4.x")
        n, c, h2, w2 = x.shape
        if h2 % 2 or w2 % 2:
            raise ValueError("phase_split: H and W must be even")
        ps = [torch.empty((n, c, h2 // 2, w2 // 2), device=x.device, dtype=torch.float32) for _ in range(4)]
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ps])
        _lib.check(_lib.load().pg_phase_merge4(x.data_ptr(), ptrs, n * c, h2 // 2, w2 // 2, 0, _stream()), "pg_phase_merge4")
        return tuple(ps)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes

        like = next(g for g in gs if g is not None)
        gs = [zeros_like(like) if g is None else _chk(g, "phase_split4.g") for g in gs]
        n, c, h, w = like.shape
        dx = torch.empty((n, c, 2 * h, 2 * w), device=like.device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gs])
        _lib.check(_lib.load().pg_phase_merge4(dx.data_ptr(), ptrs, n * c, h, w, 1, _stream()), "pg_phase_merge4")
        return dx


def phase_split4(x):
    """The four 2x2 phases of x as separate tensors (see _PhaseSplit4)."""
    return _PhaseSplit4.apply(x)


def phase_merge4(phases):
    """phase_merge(torch.stack(phases)) without the stacked tensor."""
    return _PhaseMerge4.apply(*phases)


# --------------------------------------------------------------------------------------------
# PixelCNN++ pieces (not in the reference)
# --------------------------------------------------------------------------------------------
class _ConcatElu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _chk(x, "concat_elu.x")
        n, c, h, w = x.shape
        y = torch.empty((n, 2 * c, h, w), device=x.device, dtype=torch.float32)
        _lib.check(lib.pg_concat_elu_fwd(x.data_ptr(), y.data_ptr(), n, c * h * w, _stream()), "pg_concat_elu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "concat_elu.dy")
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.check(lib.pg_concat_elu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c * h * w, _stream()),
                   "pg_concat_elu_bwd")
        return dx


def concat_elu(x):
    """[elu(x) | elu(-x)] along the channels."""
    return _ConcatElu.apply(x)


class _Resample2(torch.autograd.Function):
    """up=False: y = x[:, :, ::2, ::2]; up=True: y (2H, 2W) with y[:, :, ::2, ::2] = x and zeros elsewhere.
    Each is the other's adjoint; both run on pg_phase_merge4 (phase 0 of the 2x2 phase decomposition; the other three phases are
    one shared zero / dump tensor: no stacked buffer, no copy)."""

    @staticmethod
    def forward(ctx, x, up):
        ctx.up = up
        return _Resample2._run(x, up)

    @staticmethod
    def _run(x, up):
        import ctypes

        lib = _lib.load()
        x = _chk(x, "resample2.x")
        n, c, h, w = x.shape
        if up:  # phase 0 = x, the other three phases read one zero tensor
            z = zeros((n, c, h, w), x.device)
            y = torch.empty((n, c, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * 4)(x.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr())
            _lib.check(lib.pg_phase_merge4(y.data_ptr(), ptrs, n * c, h, w, 1, _stream()), "pg_phase_merge4")
            return y
        if h % 2 or w % 2:
            raise ValueError("subsample2: H and W must be even")
        y = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
        dump = torch.empty_like(y)  # the three unused phases land here (never read)
        ptrs = (ctypes.c_void_p * 4)(y.data_ptr(), dump.data_ptr(), dump.data_ptr(), dump.data_ptr())
        _lib.check(lib.pg_phase_merge4(x.data_ptr(), ptrs, n * c, h // 2, w // 2, 0, _stream()), "pg_phase_merge4")
        return y

    @staticmethod
    def backward(ctx, dy):
        return _Resample2._run(dy, not ctx.up), None


def subsample2(x):
    return _Resample2.apply(x, False)


def zero_insert2(x):
    return _Resample2.apply(x, True)
