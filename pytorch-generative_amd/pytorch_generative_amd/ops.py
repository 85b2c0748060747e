"""torch.autograd bindings of the HIP kernels (the only arithmetic on the hot path).

Every function here launches kernels from libpg_hip.so on torch's *current* HIP stream with
raw device pointers (so a whole training step can be captured into a hipGraph). PyTorch is
used for memory (caching allocator), streams and the autograd tape only.

Weight gradients: if a parameter carries a `_pg_grad` tensor (a view into the trainer's flat
gradient buffer, zeroed once per step) the backward kernels accumulate straight into it with
fp32 atomics and autograd sees `None` for that parameter — no per-parameter AccumulateGrad
kernels, and the flat buffer is what RCCL all-reduces. Without `_pg_grad` the usual
`param.grad` protocol is followed.
"""

import math

import os

import torch

from pytorch_generative_amd import _lib

ACT_NONE, ACT_RELU, ACT_ELU, ACT_GELU = 0, 1, 2, 3
ACT_ELU_OUT = 4  # dgrad epilogue only: derivative of ELU from its output (include/pg_hip.h)
GATE_TANH, GATE_IDENTITY = 0, 1
_ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "elu": ACT_ELU, "gelu": ACT_GELU}


def _stream():
    return torch.cuda.current_stream().cuda_stream


class RowDecode:
    """Context of row-cached incremental sampling (models/base.py).

    While active, the convolutional models' ordinary forward() is called on ONE image row
    (N, C, 1, W): every layer of these models is row-causal (its output row r depends on input rows
    <= r only), so nn.Conv2d keeps the last k input rows it needs (k = upward reach of its taps) in a
    private band buffer and evaluates only the current row; nn.CausalAttention keeps its q / k / v
    maps; image_positional_encoding returns the current row of the full-size encoding. `commit`
    marks the pass that runs once a row is final and pushes it into the caches."""

    current = None

    def __init__(self, height):
        self.height, self.row, self.commit = int(height), 0, False

    def __enter__(self):
        RowDecode.current = self
        return self

    def __exit__(self, *exc):
        RowDecode.current = None
        return False


def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}; "
            "the HIP operator path has no CPU fallback"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream with raw pointers: a tensor of another
        # GPU would be dereferenced from the wrong device (Trainer / recipes call set_device)
        raise RuntimeError(f"{name}: tensor lives on {t.device} but the current device is "
                           f"cuda:{torch.cuda.current_device()}; call torch.cuda.set_device first")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return 0 if t is None else t.data_ptr()


def zeros(shape, device):
    """torch.zeros(shape, dtype=float32) with the fill done by the library's own kernel (pg_fill): gradient sinks, loss scalars,
    KL accumulators — no ATen fill kernel inside a captured step."""
    t = torch.empty(shape, device=device, dtype=torch.float32)
    if t.numel():
        _lib.check(_lib.load().pg_fill(t.data_ptr(), 0.0, t.numel(), _stream()), "pg_fill")
    return t


def zeros_like(t):
    return zeros(tuple(t.shape), t.device)


def _sink(param):
    """Returns the direct gradient sink of a parameter (or None)."""
    return getattr(param, "_pg_grad", None) if param is not None else None


# --------------------------------------------------------------------------------------------
# convolution over a tap list
# --------------------------------------------------------------------------------------------
class ConvSpec:
    """Static description of one stride-1 convolution as tap lists.

    out[r, c] = sum_t w[:, :, u_t, v_t] . x[r + u_t - pad_h, c + v_t - pad_w]
    `active` restricts the forward/data-grad taps (the causal mask's non-zero entries,
    reference nn/convolution.py:35-39). The weight gradient always covers `wgrad_taps`
    (all taps by default: the reference's weight.grad is unmasked, nn/convolution.py:42).
    """

    def __init__(self, kh, kw, pad_h, pad_w, active=None, wgrad_all=True):
        self.kh, self.kw, self.pad_h, self.pad_w = kh, kw, pad_h, pad_w
        all_taps = [(u, v) for u in range(kh) for v in range(kw)]
        act = all_taps if active is None else [t for t in all_taps if t in set(active)]
        if not act:
            raise ValueError("ConvSpec: no active taps")
        if len(all_taps) > 64:
            raise ValueError(f"ConvSpec: {kh}x{kw} kernel has more than 64 taps")
        self.fwd_taps = act
        self.wg_taps = all_taps if wgrad_all else act
        ia = _lib.int_array
        self.f_dr = ia([u - pad_h for u, _ in act])
        self.f_dc = ia([v - pad_w for _, v in act])
        self.f_ndr = ia([pad_h - u for u, _ in act])
        self.f_ndc = ia([pad_w - v for _, v in act])
        self.f_u = ia([u for u, _ in act])
        self.f_v = ia([v for _, v in act])
        self.w_dr = ia([u - pad_h for u, _ in self.wg_taps])
        self.w_dc = ia([v - pad_w for _, v in self.wg_taps])
        self.w_u = ia([u for u, _ in self.wg_taps])
        self.w_v = ia([v for _, v in self.wg_taps])
        # extent of the active tap list (identical for the negated list of the data gradient)
        self.hr = max(u for u, _ in act) - min(u for u, _ in act)
        self.hc = max(v for _, v in act) - min(v for _, v in act)

    def full_out(self, h, w):
        return h + 2 * self.pad_h - self.kh + 1, w + 2 * self.pad_w - self.kw + 1


# PG_CONV_MFMA=0 keeps every convolution on the VALU tap kernels (A/B measurements)
CONV_MFMA = os.environ.get("PG_CONV_MFMA", "1") != "0"


def _use_mfma(lib, k_channels, m_channels, spec, out_hw, in_w):
    """Fragment format of the matrix-core path for this problem (0: none -> VALU tap kernels,
    1: fp32 MFMA, 2: bf16x3 MFMA; include/pg_hip.h PG_CONV_FMT_*)."""
    if not CONV_MFMA:
        return 0
    return int(lib.pg_conv_mfma_supported(k_channels, m_channels, len(spec.fwd_taps), out_hw[0],
                                          out_hw[1], in_w, spec.hr, spec.hc))


def _pack_frag(lib, weight, spec, transpose, fmt):
    """MFMA A-fragment pack of the active taps (csrc/conv_mfma.hip, csrc/conv_b3.hip)."""
    cout, cin, kh, kw = weight.shape
    kc, m = (cout, cin) if transpose else (cin, cout)
    t = len(spec.fwd_taps)
    wfrag = torch.empty(lib.pg_conv_frag_floats(kc, m, t, fmt), device=weight.device, dtype=torch.float32)
    _lib.check(
        lib.pg_pack_conv_weight_frag(weight.data_ptr(), wfrag.data_ptr(), cout, cin, kh, kw, t,
                                     spec.f_u, spec.f_v, int(transpose), fmt, _stream()),
        "pg_pack_conv_weight_frag",
    )
    return wfrag


def _pack_frag_both(lib, weight, spec, fmt_f, fmt_t):
    """Forward and data-gradient fragments (one launch when both use the fp32 format)."""
    cout, cin, kh, kw = weight.shape
    t = len(spec.fwd_taps)
    wf = torch.empty(lib.pg_conv_frag_floats(cin, cout, t, fmt_f), device=weight.device, dtype=torch.float32)
    wt = torch.empty(lib.pg_conv_frag_floats(cout, cin, t, fmt_t), device=weight.device, dtype=torch.float32)
    _lib.check(
        lib.pg_pack_conv_weight_frag2(weight.data_ptr(), wf.data_ptr(), wt.data_ptr(), cout, cin, kh,
                                      kw, t, spec.f_u, spec.f_v, fmt_f, fmt_t, _stream()),
        "pg_pack_conv_weight_frag2",
    )
    return wf, wt


def _pack(lib, weight, spec, transpose):
    cout, cin, kh, kw = weight.shape
    a, b = (cout, cin) if transpose else (cin, cout)
    t = len(spec.fwd_taps)
    b_pad = lib.pg_conv_b_pad(b)
    wpk = torch.empty(a * t * b_pad, device=weight.device, dtype=torch.float32)
    _lib.check(
        lib.pg_pack_conv_weight(
            weight.data_ptr(), wpk.data_ptr(), cout, cin, kh, kw, t, spec.f_u, spec.f_v,
            int(transpose), b_pad, _stream(),
        ),
        "pg_pack_conv_weight",
    )
    return wpk


class _ConvTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, res, spec, out_hw, in_act, gw, gb, out_act=ACT_NONE,
                out_pre_scaled=False, in_post=ACT_NONE, n_skip=0, res2=None):
        lib = _lib.load()
        x = _chk(x, "conv2d.x")
        weight = _chk(weight, "conv2d.weight")
        n, cin, ih, iw = x.shape
        cout = weight.shape[0]
        if weight.shape[1] != cin:
            raise ValueError(f"conv2d: weight expects {weight.shape[1]} input channels, got {cin}")
        oh, ow = out_hw
        if bias is not None:
            bias = _chk(bias, "conv2d.bias")
        if res is not None:
            res = _chk(res, "conv2d.res")
            if tuple(res.shape) != (n, cout, oh, ow):
                raise ValueError("conv2d: residual shape mismatch")
        out = torch.empty((n, cout, oh, ow), device=x.device, dtype=torch.float32)
        mfma = _use_mfma(lib, cin, cout, spec, (oh, ow), iw)
        if res2 is not None:
            # a channel slice of a wider tensor is read with its batch stride (bf16x3 epilogue), no copy
            res2 = res2 if _dense_per_image(res2) else _chk(res2, "conv2d.res2")
            if res is None or tuple(res2.shape) != (n, cout, oh, ow):
                raise ValueError("conv2d: res2 needs res and the output's shape")
            if not (mfma == CONV_FMT_B3 and cout >= 64):
                raise ValueError("conv2d: a second residual needs the bf16x3 kernel with >= 64 output channels "
                                 "(check ops.conv_two_residuals_ok first)")
        if out_pre_scaled and res is not None:
            # the consumer (in_post) recovers act' from THIS output: a residual added behind the activation would change
            # the value it reads and with it every gradient upstream (found by tests/test_gpu_ops.py::test_conv_protocol_matrix)
            raise ValueError("conv2d: out_pre_scaled cannot be combined with a residual (the consumer's in_post derivative "
                             "is taken from this convolution's output)")
        if out_pre_scaled and out_act == ACT_NONE:
            raise ValueError("conv2d: out_pre_scaled without an output activation")
        if (out_act != ACT_NONE or in_post != ACT_NONE) and not mfma:
            raise ValueError("conv2d: fused output activations need the matrix-core path "
                             "(check ops.conv_mfma_ok first)")
        ctx.wfrag_t = None
        if mfma:
            fmt_t = _use_mfma(lib, cout, cin, spec, (ih, iw), ow) if ctx.needs_input_grad[0] else 0
            if fmt_t:
                wfrag, ctx.wfrag_t = _pack_frag_both(lib, weight, spec, mfma, fmt_t)  # backward's fragments too
            else:
                wfrag = _pack_frag(lib, weight, spec, False, mfma)
            _lib.check(
                lib.pg_conv2d_mfma_ex(
                    x.data_ptr(), wfrag.data_ptr(), _p(bias), _p(res), out.data_ptr(), n, cin, ih,
                    iw, cout, oh, ow, len(spec.fwd_taps), spec.f_dr, spec.f_dc, in_act, 0, ACT_NONE,
                    out_act, mfma, _p(res2), 0, res2.stride(0) if res2 is not None else 0, _stream(),
                ),
                "pg_conv2d_mfma",
            )
        else:
            wpk = _pack(lib, weight, spec, transpose=False)
            _lib.check(
                lib.pg_conv2d_taps(
                    x.data_ptr(), wpk.data_ptr(), _p(bias), _p(res), out.data_ptr(), n, cin, ih, iw,
                    cout, oh, ow, len(spec.fwd_taps), spec.f_dr, spec.f_dc, in_act, 0, ACT_NONE,
                    _stream(),
                ),
                "pg_conv2d_taps",
            )
        if out_act != ACT_NONE and not out_pre_scaled:
            # backward recovers act' from the output: v = out - res
            ctx.save_for_backward(x, weight, out, res) if res is not None else ctx.save_for_backward(x, weight, out)
        else:
            ctx.save_for_backward(x, weight)
        ctx.spec, ctx.in_act, ctx.has_bias, ctx.has_res = spec, in_act, bias is not None, res is not None
        ctx.has_res2 = res2 is not None
        ctx.gw, ctx.gb = gw, gb
        ctx.out_act, ctx.out_pre_scaled, ctx.in_post = out_act, out_pre_scaled, in_post
        ctx.n_skip = n_skip
        if n_skip:
            # pass-through aliases of x for skip connections: their consumers' gradients come back to THIS
            # node's backward and are added in the data gradient's epilogue (no gradient-sum kernel)
            return (out,) + tuple(x.view_as(x) for _ in range(n_skip))
        return out

    @staticmethod
    def backward(ctx, dy, *d_skips):
        need = ctx.needs_input_grad
        return _ConvTaps.backward_impl(ctx, dy, need[0], need[1], ctx.has_bias and need[2],
                                       d_skips=d_skips) + (None, None, None, None,
                                                           dy if getattr(ctx, "has_res2", False) else None)

    @staticmethod
    def backward_impl(ctx, dy, need_dx, need_w, need_b, d_skips=()):
        lib = _lib.load()
        x, weight = ctx.saved_tensors[:2]
        spec = ctx.spec
        dy = _chk(dy, "conv2d.dy")
        dres = dy if ctx.has_res else None
        out_act = getattr(ctx, "out_act", ACT_NONE)
        in_post = getattr(ctx, "in_post", ACT_NONE)
        if out_act != ACT_NONE and not ctx.out_pre_scaled:
            out = ctx.saved_tensors[2]
            res = ctx.saved_tensors[3] if ctx.has_res else None
            g = torch.empty_like(dy)
            _lib.check(lib.pg_act_bwd_from_out(out.data_ptr(), _p(res), dy.data_ptr(), g.data_ptr(),
                                               dy.numel(), out_act, _stream()), "pg_act_bwd_from_out")
            dy = g
        n, cin, ih, iw = x.shape
        _, cout, oh, ow = dy.shape
        dx = dw = db = None
        fmt_t = _use_mfma(lib, cout, cin, spec, (ih, iw), ow) if need_dx else 0
        # the producer of x skipped its activation derivative on the promise that THIS data gradient's
        # epilogue applies it (out_pre_scaled / in_post protocol): only the matrix-core epilogue can
        if in_post != ACT_NONE and ctx.in_act != ACT_NONE:
            raise ValueError("conv2d: in_act and in_post cannot both be set (one epilogue derivative)")
        if in_post != ACT_NONE and need_dx and not fmt_t:
            raise RuntimeError("conv2d: in_post needs the matrix-core data gradient (shape not covered): "
                               "the activation derivative of the producer would be dropped")
        if fmt_t:
            # matrix-core data gradient; act'(x) of a fused input activation in its epilogue (one
            # exp / erf per output element is noise next to the MFMA work of the tile)
            wfrag_t = getattr(ctx, "wfrag_t", None)
            if wfrag_t is None:
                wfrag_t = _pack_frag(lib, weight, spec, True, fmt_t)
            dx = torch.empty_like(x)
            dact = ctx.in_act if ctx.in_act != ACT_NONE else (ACT_ELU_OUT if in_post == ACT_ELU else ACT_NONE)
            fuse = dact != ACT_NONE
            # pass-through gradients of skip connections on x: up to two ride in the epilogue of the
            # bf16x3 kernel (dx = dgrad * act' + skip1 + skip2); a skip may be a channel slice of a wider
            # gradient (batch-strided)
            skips = [g for g in d_skips if g is not None]
            fused_skips = []
            if fmt_t == CONV_FMT_B3 and cin >= 64:  # the multi-stream epilogue exists for >= 64 output channels
                while skips and len(fused_skips) < 2:
                    g = skips.pop(0)
                    fused_skips.append(g if _dense_per_image(g) else _chk(g, "conv2d.d_skip"))
            r1 = fused_skips[0] if fused_skips else None
            r2 = fused_skips[1] if len(fused_skips) > 1 else None
            _lib.check(
                lib.pg_conv2d_mfma_ex(
                    dy.data_ptr(), wfrag_t.data_ptr(), 0, _p(r1), dx.data_ptr(), n, cout, oh, ow, cin,
                    ih, iw, len(spec.fwd_taps), spec.f_ndr, spec.f_ndc, ACT_NONE,
                    x.data_ptr() if fuse else 0, dact, ACT_NONE, fmt_t, _p(r2),
                    r1.stride(0) if r1 is not None else 0, r2.stride(0) if r2 is not None else 0, _stream(),
                ),
                "pg_conv2d_mfma(dgrad)",
            )
            for g in skips:  # more than two, or not the bf16x3 format
                dx = add(dx, _chk(g, "conv2d.d_skip"))
        elif need_dx:
            wpk_t = _pack(lib, weight, spec, transpose=True)
            dx = torch.empty_like(x)
            # ReLU's derivative is applied in the dgrad kernel's epilogue; for ELU/GELU (exp/erf:
            # ~40 instructions per element) the fused epilogue measured SLOWER than a separate
            # streaming pass (107 us vs 25 + 47 us on the ImageGPT MLP), so they stay separate.
            fuse = ctx.in_act == ACT_RELU
            _lib.check(
                lib.pg_conv2d_taps(
                    dy.data_ptr(), wpk_t.data_ptr(), 0, 0, dx.data_ptr(), n, cout, oh, ow, cin, ih,
                    iw, len(spec.fwd_taps), spec.f_ndr, spec.f_ndc, ACT_NONE,
                    x.data_ptr() if fuse else 0, ctx.in_act if fuse else ACT_NONE, _stream(),
                ),
                "pg_conv2d_taps(dgrad)",
            )
            if ctx.in_act != ACT_NONE and not fuse:
                _lib.check(
                    lib.pg_act_bwd(x.data_ptr(), dx.data_ptr(), dx.data_ptr(), dx.numel(),
                                   ctx.in_act, _stream()),
                    "pg_act_bwd",
                )
            for g in d_skips:
                if g is not None:
                    dx = add(dx, _chk(g, "conv2d.d_skip"))
        if need_w or need_b:
            gw, gb = ctx.gw, ctx.gb
            if gw is None:
                dw = zeros_like(weight)
                gw_t = dw
            else:
                gw_t = gw
            if need_b:
                if gb is None:
                    db = zeros((cout,), x.device)
                    gb_t = db
                else:
                    gb_t = gb
            else:
                gb_t = None
            ws_n = lib.pg_conv2d_wgrad_workspace_floats(cout, cin, len(spec.wg_taps))
            ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
            _lib.check(
                lib.pg_conv2d_wgrad(
                    x.data_ptr(), dy.data_ptr(), gw_t.data_ptr(), _p(gb_t), n, cin, ih, iw, cout,
                    oh, ow, spec.kh, spec.kw, len(spec.wg_taps), spec.w_dr, spec.w_dc, spec.w_u,
                    spec.w_v, ctx.in_act, ws.data_ptr(), ws_n, _stream(),
                ),
                "pg_conv2d_wgrad",
            )
        if not need_dx and any(g is not None for g in d_skips):
            raise RuntimeError("conv2d: skip outputs of an input that needs no gradient received gradients")
        return dx, dw, db, dres, None, None, None, None, None


CONV_FMT_F32, CONV_FMT_B3 = 1, 2  # include/pg_hip.h PG_CONV_FMT_*
FUSE_SKIP = os.environ.get("PG_FUSE_SKIP", "1") != "0"  # A/B: 0 = plain fan-out, autograd sums the gradients


def _dense_per_image(t):
    """True if every image of the (N, C, H, W) tensor is a dense (C, H, W) block and the data is fp32 on
    the GPU: the batch stride may be larger than C*H*W (a channel slice of a wider tensor)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        return False
    _, c, h, w = t.shape
    st = t.stride()
    return st[3] == 1 and st[2] == w and st[1] == h * w and st[0] >= c * h * w and t.data_ptr() % 16 == 0


# A/B switches for measurements (PG_FUSE_PAIR=0 / PG_FUSE_LNSKIP=0 select the unfused graphs)
FUSE_PAIR = os.environ.get("PG_FUSE_PAIR", "1") != "0"
FUSE_LNSKIP = os.environ.get("PG_FUSE_LNSKIP", "1") != "0"
FUSE_QKV_EXTRA = os.environ.get("PG_FUSE_QKV_EXTRA", "1") != "0"  # CausalAttention with extra_x: merged [q|k|v] projection


def _adjacent_view(a, b, shape):
    """One tensor over `a` followed immediately by `b` in the same storage (None if they are not
    laid out that way): FlatAdam places declared pairs back to back (optim.py, `_pg_follows`)."""
    if a is None or b is None or a.dtype != torch.float32 or b.dtype != torch.float32:
        return None
    if not (a.is_contiguous() and b.is_contiguous()) or a.device != b.device:
        return None
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return None
    if a.data_ptr() + 4 * a.numel() != b.data_ptr():
        return None
    strides, acc = [], 1
    for d in reversed(shape):
        strides.append(acc)
        acc *= d
    return a.as_strided(tuple(shape), tuple(reversed(strides)), a.storage_offset())


def conv_pair_views(conv_a, conv_b):
    """(weight, bias, weight-grad sink, bias-grad sink) of the concatenated convolution
    [conv_a; conv_b] as zero-copy views, or None when the two modules' parameters / gradient sinks
    are not adjacent in the flat buffers (then the caller runs the two convolutions separately)."""
    wa, wb = conv_a.weight, conv_b.weight
    if conv_a.bias is None or conv_b.bias is None or wa.shape[1:] != wb.shape[1:]:
        return None
    ca, cb = wa.shape[0], wb.shape[0]
    wshape = (ca + cb,) + tuple(wa.shape[1:])
    w = _adjacent_view(wa.data, wb.data, wshape)
    b = _adjacent_view(conv_a.bias.data, conv_b.bias.data, (ca + cb,))
    gw = _adjacent_view(_sink(wa), _sink(wb), wshape)
    gb = _adjacent_view(_sink(conv_a.bias), _sink(conv_b.bias), (ca + cb,))
    if w is None or b is None or gw is None or gb is None:
        return None
    return w, b, gw, gb


class _ConvPair(torch.autograd.Function):
    """y = [conv_a(x); conv_b(x)] (channel concatenation) as ONE tap convolution over the merged
    parameter views: x is read once, one data gradient (no autograd accumulation pass over dx), one
    weight-gradient launch writing straight into the merged gradient sinks."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, views, spec, out_hw):
        w, b, gw, gb = views
        out = _ConvTaps.forward(ctx, x, w, b, None, spec, out_hw, ACT_NONE, gw, gb)
        return out

    @staticmethod
    def backward(ctx, dy):
        need = ctx.needs_input_grad
        dx = _ConvTaps.backward_impl(ctx, dy, need[0], need[1] or need[3], need[2] or need[4])[0]
        return dx, None, None, None, None, None, None, None


def conv2d_pair(x, conv_a, conv_b, views, spec, out_hw=None):
    if out_hw is None:
        out_hw = spec.full_out(x.shape[2], x.shape[3])
    return _ConvPair.apply(x, conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias, views, spec,
                           tuple(out_hw))


def conv_mfma_ok(x, weight, spec, out_hw=None):
    """True if this convolution AND its data gradient run on the matrix-core kernels (the fused
    output-activation / post-activation-input protocols of conv2d_taps need both)."""
    if out_hw is None:
        out_hw = spec.full_out(x.shape[2], x.shape[3])
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    return bool(x.is_cuda and _use_mfma(lib, cin, cout, spec, out_hw, x.shape[3])
                and _use_mfma(lib, cout, cin, spec, (x.shape[2], x.shape[3]), out_hw[1]))


def conv_two_residuals_ok(x, weight, spec, out_hw=None):
    """True if conv2d_taps(..., res=, res2=) is available for this problem (bf16x3 kernel, >= 64 output channels)."""
    if out_hw is None:
        out_hw = spec.full_out(x.shape[2], x.shape[3])
    cout, cin = weight.shape[0], weight.shape[1]
    return bool(x.is_cuda and cout >= 64
                and _use_mfma(_lib.load(), cin, cout, spec, out_hw, x.shape[3]) == CONV_FMT_B3)


def conv2d_taps(x, weight, bias, spec, out_hw=None, in_act=ACT_NONE, res=None,
                weight_param=None, bias_param=None, out_act=ACT_NONE, out_pre_scaled=False,
                in_post=ACT_NONE, n_skip=0, res2=None):
    """y = out_act(conv(in_act(x)) + bias) (+ res), cropped to out_hw (defaults to the full extent).

    out_act (matrix-core path only): activation fused into the epilogue; its backward recovers act'
    from the output (ELU / ReLU). out_pre_scaled=True declares that the ONLY consumer of y hands back
    a gradient already multiplied by act'(y) — the consumer is a convolution called with
    in_post=<that activation>, which applies the factor in its data-gradient epilogue.

    n_skip > 0 returns (y, x_1, ..., x_n): pass-through aliases of x for the skip connections that also
    read x (residual adds, later concatenations). Using them instead of x makes this op x's ONLY consumer,
    so autograd never sums gradients for x: the skip gradients arrive in this op's backward and are added
    in the data-gradient kernel's epilogue."""
    if out_hw is None:
        out_hw = spec.full_out(x.shape[2], x.shape[3])
    # outputs beyond the "full" extent read only zero padding; allow up to one kernel's worth
    # (used by the phase-decomposed stride-2 convolutions)
    full = spec.full_out(x.shape[2], x.shape[3])
    if out_hw[0] > full[0] + spec.kh or out_hw[1] > full[1] + spec.kw or min(out_hw) < 1:
        raise ValueError(f"conv2d: requested output {out_hw} exceeds the full extent {full}")
    if n_skip and not FUSE_SKIP:
        y = _ConvTaps.apply(x, weight, bias, res, spec, tuple(out_hw), in_act, _sink(weight_param),
                            _sink(bias_param), out_act, bool(out_pre_scaled), in_post, 0, res2)
        return (y,) + (x,) * int(n_skip)
    return _ConvTaps.apply(x, weight, bias, res, spec, tuple(out_hw), in_act,
                           _sink(weight_param), _sink(bias_param), out_act, bool(out_pre_scaled),
                           in_post, int(n_skip), res2)


# --------------------------------------------------------------------------------------------
# fused position-wise MLP (Conv2d 1x1 -> GELU -> Conv2d 1x1 [+ residual])
# --------------------------------------------------------------------------------------------
FUSE_MLP = os.environ.get("PG_FUSE_MLP", "1") != "0"


def mlp_gelu_supported(x, conv1, conv2):
    """The fused kernels are instantiated for the ImageGPT block shape (C = 16 -> 64 -> 16, 1x1,
    L % 16 == 0); anything else runs as conv -> gelu -> conv."""
    if not FUSE_MLP or conv1.bias is None or conv2.bias is None:
        return False
    w1, w2 = conv1.weight, conv2.weight
    return (tuple(w1.shape) == (64, 16, 1, 1) and tuple(w2.shape) == (16, 64, 1, 1)
            and x.shape[1] == 16 and (x.shape[2] * x.shape[3]) % 16 == 0)


class _MlpGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, sinks):
        lib = _lib.load()
        x = _chk(x, "mlp_gelu.x")
        w1, b1, w2, b2 = (_chk(t, "mlp_gelu.param") for t in (w1, b1, w2, b2))
        if res is not None:
            res = _chk(res, "mlp_gelu.res")
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        _lib.check(
            lib.pg_mlp_gelu_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                b2.data_ptr(), _p(res), y.data_ptr(), n, c, w1.shape[0], h * w,
                                _stream()),
            "pg_mlp_gelu_fwd",
        )
        ctx.save_for_backward(x, w1, b1, w2)
        ctx.sinks, ctx.has_res = sinks, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w1, b1, w2 = ctx.saved_tensors
        dy = _chk(dy, "mlp_gelu.dy")
        n, c, h, w = x.shape
        hd = w1.shape[0]
        dx = torch.empty_like(x)
        grads, outs = [], []
        for sink, like in zip(ctx.sinks, (w1, b1, w2, None)):
            if sink is not None:
                grads.append(sink)
                outs.append(None)
            else:
                t = zeros((c,), x.device) if like is None else zeros_like(like)
                grads.append(t)
                outs.append(t)
        ws_n = lib.pg_mlp_gelu_bwd_workspace_floats(n, h * w)
        ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
        _lib.check(
            lib.pg_mlp_gelu_bwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                dy.data_ptr(), dx.data_ptr(), grads[0].data_ptr(),
                                grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), n, c,
                                hd, h * w, ws.data_ptr(), ws_n, _stream()),
            "pg_mlp_gelu_bwd",
        )
        return dx, outs[0], outs[1], outs[2], outs[3], (dy if ctx.has_res else None), None


def mlp_gelu(x, conv1, conv2, res=None):
    """res + conv2(gelu(conv1(x))) for two 1x1 convolutions, hidden activations kept in registers
    (check mlp_gelu_supported first)."""
    sinks = (_sink(conv1.weight), _sink(conv1.bias), _sink(conv2.weight), _sink(conv2.bias))
    return _MlpGelu.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias, res, sinks)


# --------------------------------------------------------------------------------------------
# ImageGPT transformer block minus the attention core: fused head / tail (gpt_block.hip)
# --------------------------------------------------------------------------------------------
FUSE_BLOCK = os.environ.get("PG_FUSE_BLOCK", "1") != "0"
DEFER_BLOCK_REDUCE = os.environ.get("PG_BLOCK_REDUCE_MERGED", "1") != "0"  # A/B: 0 = two reduce launches per block


def _grad_targets(params):
    """Per parameter: (tensor the kernel adds into, value to return to autograd): the direct sink if
    the parameter has one (then autograd gets None), else a fresh zero tensor."""
    tgt, ret = [], []
    for p in params:
        sink = _sink(p)
        if sink is not None:
            tgt.append(sink)
            ret.append(None)
        else:
            z = zeros_like(p)
            tgt.append(z)
            ret.append(z)
    return tgt, ret


class _GPTBlockHead(torch.autograd.Function):
    """(qkv, x) = ([W_q; W_kv] LN1(x) + b, x). The second output aliases x: whatever gradient reaches it
    (the residual routes of the block) is added to LN1's input gradient inside the backward kernel."""

    @staticmethod
    def forward(ctx, x, lnw, lnb, wq, bq, wkv, bkv, eps, params, pair=None):
        lib = _lib.load()
        ctx.pair = pair
        x_in = x
        x = _chk(x, "gpt_block_head.x")
        n, c, h, w = x.shape
        qkv = torch.empty((n, 3 * c, h, w), device=x.device, dtype=torch.float32)
        _lib.check(
            lib.pg_gpt_block_head_fwd(x.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), wq.data_ptr(),
                                      bq.data_ptr(), wkv.data_ptr(), bkv.data_ptr(), qkv.data_ptr(),
                                      n, c, h * w, eps, _stream()),
            "pg_gpt_block_head_fwd",
        )
        ctx.save_for_backward(x, lnw, lnb, wq, wkv)
        ctx.eps, ctx.params = eps, params
        return qkv, x_in

    @staticmethod
    def backward(ctx, dqkv, gx):
        lib = _lib.load()
        x, lnw, lnb, wq, wkv = ctx.saved_tensors
        n, c, h, w = x.shape
        pending = ctx.pair.pop("tail", None) if ctx.pair is not None else None
        if dqkv is None:
            if pending is not None:
                raise RuntimeError("gpt_block_head: a deferred tail reduction is pending but the head has no gradient")
            return gx, None, None, None, None, None, None, None, None, None
        dqkv = _chk(dqkv, "gpt_block_head.dqkv")
        gx = zeros_like(x) if gx is None else _chk(gx, "gpt_block_head.gx")
        dx = torch.empty_like(x)
        tgt, ret = _grad_targets(ctx.params)  # order: lnw, lnb, wq, bq, wkv, bkv
        ws_n = lib.pg_gpt_block_head_bwd_workspace_floats(n, h * w)
        ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
        head_args = (x.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), wq.data_ptr(),
                     wkv.data_ptr(), dqkv.data_ptr(), gx.data_ptr(), dx.data_ptr(),
                     tgt[0].data_ptr(), tgt[1].data_ptr(), tgt[2].data_ptr(),
                     tgt[3].data_ptr(), tgt[4].data_ptr(), tgt[5].data_ptr(), n, c,
                     h * w, ctx.eps, ws.data_ptr(), ws_n)
        if pending is not None:  # this block's tail kernel left its partial rows: ONE reduce launch for both
            t_ws, t = pending    # t order: wp, bp, lnw, lnb, w1, b1, w2, b2
            _lib.check(
                lib.pg_gpt_block_head_bwd_with_tail(*head_args, t_ws.data_ptr(), t[4].data_ptr(), t[5].data_ptr(),
                                                    t[6].data_ptr(), t[7].data_ptr(), t[0].data_ptr(),
                                                    t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), _stream()),
                "pg_gpt_block_head_bwd_with_tail",
            )
        else:
            _lib.check(lib.pg_gpt_block_head_bwd(*head_args, _stream()), "pg_gpt_block_head_bwd")
        return (dx, *ret, None, None, None)


class _GPTBlockTail(torch.autograd.Function):
    """x_new = x + x_mid + mlp(LN2(x_mid)), x_mid = x + W_p o + b_p (projection, both residuals of the
    block and the model loop's `x + block(x)`)."""

    @staticmethod
    def forward(ctx, o, x, wp, bp, lnw, lnb, w1, b1, w2, b2, eps, params, pair=None):
        lib = _lib.load()
        ctx.pair = pair
        o = _chk(o, "gpt_block_tail.o")
        x = _chk(x, "gpt_block_tail.x")
        n, c, h, w = x.shape
        x_new = torch.empty_like(x)
        _lib.check(
            lib.pg_gpt_block_tail_fwd(o.data_ptr(), x.data_ptr(), wp.data_ptr(), bp.data_ptr(),
                                      lnw.data_ptr(), lnb.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                      w2.data_ptr(), b2.data_ptr(), x_new.data_ptr(), n, c,
                                      w1.shape[0], h * w, eps, _stream()),
            "pg_gpt_block_tail_fwd",
        )
        ctx.save_for_backward(o, x, wp, bp, lnw, lnb, w1, b1, w2)
        ctx.eps, ctx.params = eps, params
        return x_new

    @staticmethod
    def backward(ctx, d):
        lib = _lib.load()
        o, x, wp, bp, lnw, lnb, w1, b1, w2 = ctx.saved_tensors
        d = _chk(d, "gpt_block_tail.dx_new")
        n, c, h, w = x.shape
        d_o, gx = torch.empty_like(o), torch.empty_like(x)
        tgt, ret = _grad_targets(ctx.params)  # order: wp, bp, lnw, lnb, w1, b1, w2, b2
        ws_n = lib.pg_gpt_block_tail_bwd_workspace_floats(n, h * w)
        ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
        if ctx.pair is not None and DEFER_BLOCK_REDUCE and all(r is None for r in ret):
            # every gradient goes into a sink (FlatAdam): leave the partial rows for the head's backward of the
            # same block, which reduces both kernels' rows in one launch
            _lib.check(
                lib.pg_gpt_block_tail_bwd_partial(o.data_ptr(), x.data_ptr(), wp.data_ptr(), bp.data_ptr(),
                                                  lnw.data_ptr(), lnb.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                                  w2.data_ptr(), d.data_ptr(), d_o.data_ptr(), gx.data_ptr(), n, c,
                                                  w1.shape[0], h * w, ctx.eps, ws.data_ptr(), ws_n, _stream()),
                "pg_gpt_block_tail_bwd_partial",
            )
            ctx.pair["tail"] = (ws, tgt)
            return (d_o, gx, *ret, None, None, None)
        _lib.check(
            lib.pg_gpt_block_tail_bwd(o.data_ptr(), x.data_ptr(), wp.data_ptr(), bp.data_ptr(),
                                      lnw.data_ptr(), lnb.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                      w2.data_ptr(), d.data_ptr(), d_o.data_ptr(), gx.data_ptr(),
                                      tgt[0].data_ptr(), tgt[1].data_ptr(), tgt[2].data_ptr(),
                                      tgt[3].data_ptr(), tgt[4].data_ptr(), tgt[5].data_ptr(),
                                      tgt[6].data_ptr(), tgt[7].data_ptr(), n, c, w1.shape[0], h * w,
                                      ctx.eps, ws.data_ptr(), ws_n, _stream()),
            "pg_gpt_block_tail_bwd",
        )
        return (d_o, gx, *ret, None, None, None)


def gpt_block_supported(x, ln1, q, kv, proj, ln2, fc1, fc2):
    """The fused block kernels cover the BASELINE.json ImageGPT block: 16 channels, 1x1 projections
    with biases, 64 hidden units, L % 16 == 0."""
    if not FUSE_BLOCK or x.shape[1] != 16 or (x.shape[2] * x.shape[3]) % 16 != 0:
        return False
    shapes = (tuple(q.weight.shape), tuple(kv.weight.shape), tuple(proj.weight.shape),
              tuple(fc1.weight.shape), tuple(fc2.weight.shape))
    if shapes != ((16, 16, 1, 1), (32, 16, 1, 1), (16, 16, 1, 1), (64, 16, 1, 1), (16, 64, 1, 1)):
        return False
    if any(m.bias is None for m in (q, kv, proj, fc1, fc2)):
        return False
    return tuple(ln1.normalized_shape) == (16,) and tuple(ln2.normalized_shape) == (16,)


def gpt_block_head(x, ln1, q, kv, pair=None):
    """pair: a dict shared with gpt_block_tail of the SAME block (one per forward): lets the two backward
    kernels share one weight-gradient reduction launch."""
    params = (ln1.weight, ln1.bias, q.weight, q.bias, kv.weight, kv.bias)
    return _GPTBlockHead.apply(x, *params, float(ln1.eps), params, pair)


def gpt_block_tail(o, x, proj, ln2, fc1, fc2, pair=None):
    params = (proj.weight, proj.bias, ln2.weight, ln2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    return _GPTBlockTail.apply(o, x, *params, float(ln2.eps), params, pair)


# --------------------------------------------------------------------------------------------
# NCHW LayerNorm
# --------------------------------------------------------------------------------------------
class _NCHWLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, gg, gb, with_skip=False):
        lib = _lib.load()
        x_in = x
        x = _chk(x, "layernorm.x")
        n, c, h, w = x.shape
        if gamma.numel() != c:
            raise ValueError(f"NCHWLayerNorm: normalized_shape {gamma.numel()} != channels {c}")
        y = torch.empty_like(x)
        mean = torch.empty(n * h * w, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        _lib.check(
            lib.pg_nchw_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                      mean.data_ptr(), rstd.data_ptr(), n, c, h * w, eps, _stream()),
            "pg_nchw_layernorm_fwd",
        )
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.gg, ctx.gb = gg, gb
        if with_skip:
            # second output: x itself (autograd aliases it). Whatever gradient reaches x through this
            # alias — the residual branch of `x + f(LN(x))` — is added in the backward kernel's
            # epilogue instead of by a separate accumulation pass over the activation.
            return y, x_in
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        lib = _lib.load()
        x, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the skip output was used
            return dskip, None, None, None, None, None, None
        dy = _chk(dy, "layernorm.dy")
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        dg = db = None
        gg, gb = ctx.gg, ctx.gb
        if gg is None:
            dg = zeros((c,), x.device)
            gg = dg
        if gb is None:
            db = zeros((c,), x.device)
            gb = db
        ws_n = lib.pg_nchw_layernorm_bwd_workspace_floats(n, c, h * w)
        ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
        if dskip is None:
            _lib.check(
                lib.pg_nchw_layernorm_bwd(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), gg.data_ptr(),
                                          gb.data_ptr(), n, c, h * w, ws.data_ptr(), ws_n, _stream()),
                "pg_nchw_layernorm_bwd",
            )
        else:
            dskip = _chk(dskip, "layernorm.dskip")
            _lib.check(
                lib.pg_nchw_layernorm_bwd_res(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                              rstd.data_ptr(), dy.data_ptr(), dskip.data_ptr(),
                                              dx.data_ptr(), gg.data_ptr(), gb.data_ptr(), n, c,
                                              h * w, ws.data_ptr(), ws_n, _stream()),
                "pg_nchw_layernorm_bwd_res",
            )
        return dx, dg, db, None, None, None, None


def nchw_layernorm(x, weight, bias, eps=1e-5):
    return _NCHWLayerNorm.apply(x, weight, bias, float(eps), _sink(weight), _sink(bias))


def nchw_layernorm_skip(x, weight, bias, eps=1e-5):
    """(LN(x), x): use the second output for the residual branch of `x + f(LN(x))`; its gradient is
    then added to LN's input gradient inside the backward kernel."""
    return _NCHWLayerNorm.apply(x, weight, bias, float(eps), _sink(weight), _sink(bias), True)


# --------------------------------------------------------------------------------------------
# causal attention core
# --------------------------------------------------------------------------------------------
class _CausalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, n_heads, embed, vdim, strict):
        lib = _lib.load()
        q = _chk(q, "attention.q")
        kv = _chk(kv, "attention.kv")
        n, e, h, w = q.shape
        if e != embed or kv.shape[1] != embed + vdim:
            raise ValueError("attention: channel mismatch between q / kv and embed / value dims")
        if embed % n_heads or vdim % n_heads:
            raise ValueError("attention: channels not divisible by n_heads")
        L = h * w
        dk, dv = embed // n_heads, vdim // n_heads
        o = torch.empty((n, vdim, h, w), device=q.device, dtype=torch.float32)
        lse2 = torch.empty((n, n_heads, L), device=q.device, dtype=torch.float32)
        k_ptr = kv.data_ptr()
        v_ptr = kv.data_ptr() + 4 * embed * L
        _lib.check(
            lib.pg_causal_attn_fwd(q.data_ptr(), k_ptr, v_ptr, o.data_ptr(), lse2.data_ptr(), n,
                                   n_heads, L, dk, dv, embed * L, (embed + vdim) * L,
                                   (embed + vdim) * L, vdim * L, int(strict), _stream()),
            "pg_causal_attn_fwd",
        )
        ctx.save_for_backward(q, kv, o, lse2)
        ctx.cfg = (n_heads, embed, vdim, int(strict))
        return o

    @staticmethod
    def backward(ctx, d_o):
        lib = _lib.load()
        q, kv, o, lse2 = ctx.saved_tensors
        n_heads, embed, vdim, strict = ctx.cfg
        d_o = _chk(d_o, "attention.d_o")
        n, _, h, w = q.shape
        L = h * w
        dk, dv = embed // n_heads, vdim // n_heads
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        delta = torch.empty_like(lse2)
        kvs = (embed + vdim) * L
        _lib.check(
            lib.pg_causal_attn_bwd(
                q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * embed * L, o.data_ptr(),
                d_o.data_ptr(), lse2.data_ptr(), delta.data_ptr(), dq.data_ptr(), dkv.data_ptr(),
                dkv.data_ptr() + 4 * embed * L, n, n_heads, L, dk, dv, embed * L, kvs, kvs,
                vdim * L, vdim * L, embed * L, kvs, kvs, strict, _stream(),
            ),
            "pg_causal_attn_bwd",
        )
        return dq, dkv, None, None, None, None


class _CausalAttentionQKV(torch.autograd.Function):
    """Same core on ONE (N, embed + embed + vdim, H, W) tensor [q | k | v] — the output of the merged
    q/kv projection (conv2d_pair); its gradient is produced as one tensor of the same layout."""

    @staticmethod
    def forward(ctx, qkv, n_heads, embed, vdim, strict):
        lib = _lib.load()
        qkv = _chk(qkv, "attention.qkv")
        n, ch, h, w = qkv.shape
        if ch != 2 * embed + vdim:
            raise ValueError("attention: qkv channel count != 2 * embed + value dims")
        if embed % n_heads or vdim % n_heads:
            raise ValueError("attention: channels not divisible by n_heads")
        L = h * w
        dk, dv = embed // n_heads, vdim // n_heads
        o = torch.empty((n, vdim, h, w), device=qkv.device, dtype=torch.float32)
        lse2 = torch.empty((n, n_heads, L), device=qkv.device, dtype=torch.float32)
        base, bs = qkv.data_ptr(), ch * L
        _lib.check(
            lib.pg_causal_attn_fwd(base, base + 4 * embed * L, base + 8 * embed * L, o.data_ptr(),
                                   lse2.data_ptr(), n, n_heads, L, dk, dv, bs, bs, bs, vdim * L,
                                   int(strict), _stream()),
            "pg_causal_attn_fwd",
        )
        ctx.save_for_backward(qkv, o, lse2)
        ctx.cfg = (n_heads, embed, vdim, int(strict))
        return o

    @staticmethod
    def backward(ctx, d_o):
        lib = _lib.load()
        qkv, o, lse2 = ctx.saved_tensors
        n_heads, embed, vdim, strict = ctx.cfg
        d_o = _chk(d_o, "attention.d_o")
        n, ch, h, w = qkv.shape
        L = h * w
        dk, dv = embed // n_heads, vdim // n_heads
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse2)
        base, gbase, bs = qkv.data_ptr(), dqkv.data_ptr(), ch * L
        _lib.check(
            lib.pg_causal_attn_bwd(
                base, base + 4 * embed * L, base + 8 * embed * L, o.data_ptr(), d_o.data_ptr(),
                lse2.data_ptr(), delta.data_ptr(), gbase, gbase + 4 * embed * L,
                gbase + 8 * embed * L, n, n_heads, L, dk, dv, bs, bs, bs, vdim * L, vdim * L, bs, bs,
                bs, strict, _stream(),
            ),
            "pg_causal_attn_bwd",
        )
        return dqkv, None, None, None, None


class _ConcatChannels(torch.autograd.Function):
    """torch.cat(tensors, dim=1) of (N, C_i, H, W) tensors on pg_copy_rows; the gradients handed back are
    channel-slice VIEWS of the incoming gradient (no copies): the convolution they flow into adds them in
    its data-gradient epilogue (n_skip protocol) or reads them with their batch stride."""

    @staticmethod
    def forward(ctx, *tensors):
        lib = _lib.load()
        parts = [_chk(t, "concat.part") for t in tensors]
        n, _, h, w = parts[0].shape
        L = h * w
        ctot = sum(int(t.shape[1]) for t in parts)
        out = torch.empty((n, ctot, h, w), device=parts[0].device, dtype=torch.float32)
        off = 0
        for t in parts:
            if tuple(t.shape[0:1] + t.shape[2:]) != (n, h, w):
                raise ValueError("concat_channels: batch / spatial shape mismatch")
            c = int(t.shape[1])
            _lib.check(lib.pg_copy_rows(t.data_ptr(), out.data_ptr() + 4 * off * L, n, c * L, c * L, ctot * L, 0,
                                        _stream()), "pg_copy_rows")
            off += c
        ctx.sizes = [int(t.shape[1]) for t in parts]
        return out

    @staticmethod
    def backward(ctx, dy):
        grads, off = [], 0
        for i, c in enumerate(ctx.sizes):
            grads.append(dy.narrow(1, off, c) if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(grads)


def concat_channels(tensors):
    return _ConcatChannels.apply(*tensors)


class _MergeQKVWeight(torch.autograd.Function):
    """Merged weight / bias of the [q | k | v] projection over cat(x, extra_x) (nn/attention.py:139-143):
    q's rows (E, Cq) zero-padded to Ckv columns on top of kv's rows (E + V, Ckv). Backward adds the two
    row blocks of the merged gradient straight into the parameters' gradient sinks (or returns them)."""

    @staticmethod
    def forward(ctx, wq, bq, wkv, bkv, sinks):
        lib = _lib.load()
        eq, cq = int(wq.shape[0]), int(wq.shape[1])
        ekv, ckv = int(wkv.shape[0]), int(wkv.shape[1])
        w = zeros((eq + ekv, ckv, 1, 1), wq.device)
        b = torch.empty(eq + ekv, device=wq.device, dtype=torch.float32)
        st = _stream()
        _lib.check(lib.pg_copy_rows(wq.data_ptr(), w.data_ptr(), eq, cq, cq, ckv, 0, st), "pg_copy_rows")
        _lib.check(lib.pg_copy_rows(wkv.data_ptr(), w.data_ptr() + 4 * eq * ckv, 1, ekv * ckv, ekv * ckv,
                                    ekv * ckv, 0, st), "pg_copy_rows")
        _lib.check(lib.pg_copy_rows(bq.data_ptr(), b.data_ptr(), 1, eq, eq, eq, 0, st), "pg_copy_rows")
        _lib.check(lib.pg_copy_rows(bkv.data_ptr(), b.data_ptr() + 4 * eq, 1, ekv, ekv, ekv, 0, st), "pg_copy_rows")
        ctx.dims, ctx.sinks = (eq, cq, ekv, ckv), sinks
        return w, b

    @staticmethod
    def backward(ctx, dw, db):
        lib = _lib.load()
        eq, cq, ekv, ckv = ctx.dims
        dw, db = _chk(dw, "qkv.dw"), _chk(db, "qkv.db")
        st = _stream()
        outs = []
        for i, (src, rows, rl, sstride) in enumerate((
                (dw.data_ptr(), eq, cq, ckv), (db.data_ptr(), 1, eq, eq),
                (dw.data_ptr() + 4 * eq * ckv, 1, ekv * ckv, ekv * ckv), (db.data_ptr() + 4 * eq, 1, ekv, ekv))):
            sink = ctx.sinks[i]
            if sink is not None:  # accumulate into the flat gradient buffer (FlatAdam protocol)
                _lib.check(lib.pg_copy_rows(src, sink.data_ptr(), rows, rl, sstride, rl, 1, st), "pg_copy_rows")
                outs.append(None)
            else:
                g = torch.empty(rows * rl, device=dw.device, dtype=torch.float32)
                _lib.check(lib.pg_copy_rows(src, g.data_ptr(), rows, rl, sstride, rl, 0, st), "pg_copy_rows")
                outs.append(g)
        gwq, gbq, gwkv, gbkv = outs
        return (None if gwq is None else gwq.view(eq, cq, 1, 1), gbq,
                None if gwkv is None else gwkv.view(ekv, ckv, 1, 1), gbkv, None)


def merge_qkv_weight(q_conv, kv_conv):
    sinks = (_sink(q_conv.weight), _sink(q_conv.bias), _sink(kv_conv.weight), _sink(kv_conv.bias))
    return _MergeQKVWeight.apply(q_conv.weight, q_conv.bias, kv_conv.weight, kv_conv.bias, sinks)


def set_deterministic(on=True):
    """Bit-reproducible gradients (on) or the fastest kernels (off, the default). The only kernels whose
    result depends on timing are the fused attention backwards for d_k = 4 (d_v = 4: ImageGPT; d_v = 16 / 32:
    PixelSNAIL, round 4) — dQ is summed over key blocks in arrival order: last-bit differences run to run;
    `on` selects the two-kernel backward, and the per-sample KL sums of the Gaussian heads are then reduced by one
    workgroup per sample (fixed order). Still summed with fp32 atomics in arrival order, because they feed no gradient
    and no parameter: the scalar loss values (BCE / DMOL / VQ) and the squared gradient norm that `FlatAdam` reports
    (it only scales the step above max_norm = 1e50).
    Returns the previous setting."""
    prev = _lib.load().pg_attn_fused_bwd(0 if on else 1)
    return prev == 0


def causal_attention_qkv(qkv, n_heads, embed_channels, value_channels, mask_center):
    """causal_attention on the merged [q | k | v] tensor."""
    L = qkv.shape[2] * qkv.shape[3]
    if n_heads and embed_channels % n_heads == 0 and value_channels % n_heads == 0 and not attention_dims_native(
            embed_channels // n_heads, value_channels // n_heads, L):
        # head dims / L the kernels do not instantiate: the padded route of causal_attention on the two halves
        return causal_attention(qkv[:, :embed_channels].contiguous(), qkv[:, embed_channels:].contiguous(), n_heads,
                                embed_channels, value_channels, mask_center)
    return _CausalAttentionQKV.apply(qkv, n_heads, embed_channels, value_channels, bool(mask_center))


def attention_dims_native(dk, dv, L):
    """True if the kernels take these head dims / sequence length as they are (csrc/attention*.hip): d_k = d_v = 4
    and the small VALU shapes at any L, everything else as multiples of 16 with L % 16 == 0."""
    if (dk <= 4 and dv <= 32) or (dk <= 16 and dv <= 16):
        return True
    return dk in (4, 16, 32, 64) and dv in (16, 32, 64) and L % 16 == 0 and L >= 16


def _pad16(d):
    for v in (16, 32, 64):
        if d <= v:
            return v
    raise ValueError(f"attention: head dim {d} > 64 is not supported")


def causal_attention(q, kv, n_heads, embed_channels, value_channels, mask_center):
    """softmax(mask(q k^T / sqrt(d_k))) v over raster-ordered pixels; kv = cat(k, v) on dim 1.

    Head dims / sequence lengths the kernels do not instantiate (attention_dims_native) are ZERO-PADDED to the next
    instantiated size — exact: padded channels add nothing to q.k and come out of P.V as zeros, padded positions lie
    behind every real query (the causal mask hides them) and their own rows are dropped; the 1 / sqrt(d_k) of the
    REAL d_k is folded into q. The copies are ATen plumbing around the same HIP kernels (no model of the reference
    takes this route: PixelSNAIL is 4 / 32, ImageGPT 4 / 4 or 32 / 32)."""
    n, e, h, w = q.shape
    L = h * w
    if embed_channels % n_heads or value_channels % n_heads:
        raise ValueError("attention: channels not divisible by n_heads")
    dk, dv = embed_channels // n_heads, value_channels // n_heads
    if attention_dims_native(dk, dv, L):
        return _CausalAttention.apply(q, kv, n_heads, embed_channels, value_channels, bool(mask_center))
    dk_p = 4 if dk <= 4 else _pad16(dk)
    dv_p = _pad16(dv)
    L_p = -(-L // 16) * 16
    pad = torch.nn.functional.pad
    scale = (dk_p / dk) ** 0.5  # the kernel divides by sqrt(dk_p)
    q4 = pad(q.reshape(n, n_heads, dk, L) * scale, (0, L_p - L, 0, dk_p - dk))
    k4 = pad(kv[:, :embed_channels].reshape(n, n_heads, dk, L), (0, L_p - L, 0, dk_p - dk))
    v4 = pad(kv[:, embed_channels:].reshape(n, n_heads, dv, L), (0, L_p - L, 0, dv_p - dv))
    kv_p = torch.cat((k4.reshape(n, n_heads * dk_p, 1, L_p), v4.reshape(n, n_heads * dv_p, 1, L_p)), dim=1)
    o_p = _CausalAttention.apply(q4.reshape(n, n_heads * dk_p, 1, L_p).contiguous(), kv_p, n_heads,
                                 n_heads * dk_p, n_heads * dv_p, bool(mask_center))
    o = o_p.reshape(n, n_heads, dv_p, L_p)[:, :, :dv, :L]
    return o.reshape(n, value_channels, h, w)


# --------------------------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        lib = _lib.load()
        x = _chk(x, "act.x")
        y = torch.empty_like(x)
        _lib.check(lib.pg_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), act, _stream()), "pg_act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "act.dy")
        dx = torch.empty_like(x)
        _lib.check(lib.pg_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.act,
                                  _stream()), "pg_act_bwd")
        return dx, None


def relu(x):
    return _Act.apply(x, ACT_RELU)


def elu(x):
    return _Act.apply(x, ACT_ELU)


def gelu(x):
    return _Act.apply(x, ACT_GELU)


class _Gated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate, res=None):
        lib = _lib.load()
        x = _chk(x, "gated.x")
        n, c2, h, w = x.shape
        assert c2 % 2 == 0, "x must have an even number of channels."
        y = torch.empty((n, c2 // 2, h, w), device=x.device, dtype=torch.float32)
        if res is None:
            _lib.check(lib.pg_gated_fwd(x.data_ptr(), y.data_ptr(), n, c2 // 2, h * w, gate, _stream()),
                       "pg_gated_fwd")
        else:
            res = _chk(res, "gated.res")
            if res.shape != y.shape:
                raise ValueError("gated_activation: residual shape mismatch")
            _lib.check(lib.pg_gated_fwd_res(x.data_ptr(), res.data_ptr(), y.data_ptr(), n, c2 // 2,
                                            h * w, gate, _stream()), "pg_gated_fwd_res")
        ctx.save_for_backward(x)
        ctx.gate, ctx.has_res = gate, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "gated.dy")
        n, c2, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.check(lib.pg_gated_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c2 // 2, h * w,
                                    ctx.gate, _stream()), "pg_gated_bwd")
        return dx, None, (dy if ctx.has_res else None)


def gated_activation(x, gate, res=None):
    """act(x[:, :C]) * sigmoid(x[:, C:]) (+ res): GatedActivation, optionally fused with the
    residual add that follows it in PixelSNAIL's ResidualBlock (pixel_snail.py:55-56)."""
    if res is not None and ((x.shape[2] * x.shape[3]) % 4 != 0):
        return add(res, _Gated.apply(x, gate))
    return _Gated.apply(x, gate, res)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a = _chk(a, "add.a")
        b = _chk(b, "add.b")
        if a.shape != b.shape:
            raise ValueError(f"add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        _lib.check(lib.pg_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "pg_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _AddBcast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, gp):
        lib = _lib.load()
        x = _chk(x, "add_bcast.x")
        p = _chk(p, "add_bcast.p")
        per = p.numel()
        if x.numel() % per or tuple(x.shape[1:]) != tuple(p.shape[-(x.dim() - 1):]):
            raise ValueError(f"add_bcast: {tuple(p.shape)} does not broadcast over {tuple(x.shape)}")
        y = torch.empty_like(x)
        _lib.check(lib.pg_add_bcast_fwd(x.data_ptr(), p.data_ptr(), y.data_ptr(), x.shape[0], per,
                                        _stream()), "pg_add_bcast_fwd")
        ctx.gp = gp
        ctx.pshape = p.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _chk(dy, "add_bcast.dy")
        dp = None
        gp = ctx.gp
        if ctx.needs_input_grad[1]:
            if gp is None:
                dp = zeros(ctx.pshape, dy.device)
                gp = dp
            per = gp.numel()
            _lib.check(lib.pg_add_bcast_bwd(dy.data_ptr(), gp.data_ptr(), dy.shape[0], per, _stream()),
                       "pg_add_bcast_bwd")
        return (dy if ctx.needs_input_grad[0] else None), dp, None


def add_broadcast_batch(x, p):
    """x + p where p has batch dim 1 (the learned positional map of ImageGPT)."""
    return _AddBcast.apply(x, p, _sink(p))


def image_positional_encoding(shape, device):
    lib = _lib.load()
    n, _, h, w = shape
    out = torch.empty((n, 2, h, w), device=device, dtype=torch.float32)
    if not out.is_cuda:
        raise RuntimeError("image_positional_encoding: the HIP path needs a cuda device")
    _lib.check(lib.pg_image_positional_encoding(out.data_ptr(), n, h, w, _stream()),
               "pg_image_positional_encoding")
    return out


def mul_inplace_(w, mask):
    lib = _lib.load()
    if not (w.is_cuda and mask.is_cuda and w.is_contiguous() and mask.is_contiguous()):
        raise RuntimeError("mul_inplace_: expects contiguous cuda tensors")
    _lib.check(lib.pg_mul_inplace(w.data_ptr(), mask.data_ptr(), w.numel(), _stream()), "pg_mul_inplace")
    return w


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
class _BCEWithLogitsSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, x):
        lib = _lib.load()
        z = _chk(z, "bce.logits")
        x = _chk(x, "bce.targets")
        if z.numel() != x.numel():
            raise ValueError("bce: logits/targets size mismatch")
        n = z.shape[0]
        loss = zeros((1,), z.device)
        _lib.check(lib.pg_bce_logits_fwd(z.data_ptr(), x.data_ptr(), loss.data_ptr(), n,
                                         z.numel() // n, _stream()), "pg_bce_logits_fwd")
        ctx.save_for_backward(z, x)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        z, x = ctx.saved_tensors
        g = _chk(g.reshape(1), "bce.grad")
        n = z.shape[0]
        dz = torch.empty_like(z)
        _lib.check(lib.pg_bce_logits_bwd(z.data_ptr(), x.data_ptr(), g.data_ptr(), dz.data_ptr(), n,
                                         z.numel() // n, _stream()), "pg_bce_logits_bwd")
        return dz, None


class _DmolLossSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, x, n_mix):
        lib = _lib.load()
        l = _chk(l, "dmol.params")
        x = _chk(x, "dmol.images")
        n, c, h, w = l.shape
        if c != 10 * n_mix or tuple(x.shape) != (n, 3, h, w):
            raise ValueError("dmol: expected (N, 10 * n_mix, H, W) parameters and (N, 3, H, W) images in [-1, 1]")
        loss = zeros((1,), l.device)
        _lib.check(lib.pg_dmol_fwd(l.data_ptr(), x.data_ptr(), loss.data_ptr(), n, n_mix, h * w, _stream()),
                   "pg_dmol_fwd")
        ctx.save_for_backward(l, x)
        ctx.n_mix = n_mix
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        l, x = ctx.saved_tensors
        g = _chk(g.reshape(1), "dmol.grad")
        n, _, h, w = l.shape
        dl = torch.empty_like(l)
        _lib.check(lib.pg_dmol_bwd(l.data_ptr(), x.data_ptr(), g.data_ptr(), dl.data_ptr(), n, ctx.n_mix, h * w,
                                   _stream()), "pg_dmol_bwd")
        return dl, None, None


def dmol_loss_sum_mean(params, images, n_mix=10):
    """Discretized mixture-of-logistics negative log-likelihood (PixelCNN++, Salimans et al. 2017), nats,
    summed over pixels and averaged over the batch. params (N, 10 * n_mix, H, W), images (N, 3, H, W) in
    [-1, 1]. Not in the reference (BASELINE.json configs[2] names it): parity is against oracle/dmol.py."""
    return _DmolLossSumMean.apply(params, images, int(n_mix))


def bce_with_logits_sum_mean(logits, targets):
    """F.binary_cross_entropy_with_logits(reduction='none').sum(pixels).mean(batch)
    (reference image_gpt.py:158-162 and every other AR reproduce())."""
    return _BCEWithLogitsSumMean.apply(logits, targets)


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


def _sum_into(out, ts):
    """out = sum of the tensors ts (same shape as out; each dense, or dense per image with a larger batch stride) in one
    pg_sum_rows launch per 31 tensors."""
    import ctypes

    lib = _lib.load()
    n_batch, per = (int(out.shape[0]), out.numel() // max(int(out.shape[0]), 1)) if out.dim() == 4 else (1, out.numel())
    acc = None
    for i in range(0, len(ts), 31):  # 32 rows per launch, the running sum among them
        grp = ([acc] if acc is not None else []) + list(ts[i:i + 31])
        rows = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        bs = (ctypes.c_long * len(grp))(*[(int(t.stride(0)) if t.dim() == 4 else per) for t in grp])
        _lib.check(lib.pg_sum_rows(rows, bs, len(grp), out.data_ptr(), n_batch, per, _stream()), "pg_sum_rows")
        acc = out
    return out


class _SumVectors(torch.autograd.Function):
    """out = sum of k equally shaped tensors in ONE launch (pg_sum_rows); every input's gradient is the output's."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [_chk(t, "sum_vectors.t") for t in ts]
        if any(t.shape != ts[0].shape for t in ts):
            raise ValueError("sum_vectors: shape mismatch")
        ctx.k = len(ts)
        return _sum_into(torch.empty_like(ts[0]), ts)

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.k


class _Fanout(torch.autograd.Function):
    """k pass-through aliases of x for k readers: autograd then never sums gradients for x with its own chain of k - 1 `add`
    kernels — the k gradients come back HERE and are summed by one launch (pg_sum_rows), each read where it lies (a gradient
    that is a channel slice of a wider tensor, e.g. out of concat_channels' backward, with its batch stride: no copy)."""

    @staticmethod
    def forward(ctx, x, k):
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g if (g.dim() == 4 and _dense_per_image(g)) else _chk(g, "fanout.g") for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty(gs[0].shape, device=gs[0].device, dtype=torch.float32)
        return _sum_into(out, gs), None


def fanout(x, k):
    """k aliases of x, one per reader (see _Fanout); x itself when it needs no gradient or k < 2."""
    if k < 2 or not FUSE_SKIP or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(int(k), 1)
    return _Fanout.apply(x, int(k))


def sum_vectors(ts):
    """torch.stack(ts).sum(dim=0) (vd_vae.py:400, the sum of the per-block KL terms) without the stacked tensor."""
    ts = list(ts)
    if not ts:
        raise ValueError("sum_vectors: empty list")
    return ts[0] if len(ts) == 1 else _SumVectors.apply(*ts)


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


class _ElboMean(torch.autograd.Function):
    """loss = mean_n(recon_n) + mean_n(kl_n) with recon_n the per-sample BCE-with-logits sum."""

    @staticmethod
    def forward(ctx, logits, x, kl):
        lib = _lib.load()
        logits, x, kl = _chk(logits, "elbo.logits"), _chk(x, "elbo.x"), _chk(kl, "elbo.kl")
        n = logits.shape[0]
        recon = zeros((1,), logits.device)
        klm = zeros((1,), logits.device)
        _lib.check(lib.pg_bce_logits_fwd(logits.data_ptr(), x.data_ptr(), recon.data_ptr(), n,
                                         logits.numel() // n, _stream()), "pg_bce_logits_fwd")
        _lib.check(lib.pg_vec_mean_accum(kl.data_ptr(), n, klm.data_ptr(), _stream()),
                   "pg_vec_mean_accum")
        ctx.save_for_backward(logits, x)
        ctx.n = n
        return recon.view(()), klm.view(())

    @staticmethod
    def backward(ctx, g_recon, g_kl):
        lib = _lib.load()
        logits, x = ctx.saved_tensors
        n = ctx.n
        dz = torch.empty_like(logits)
        g_recon = _chk(g_recon.reshape(1), "elbo.g")
        _lib.check(lib.pg_bce_logits_bwd(logits.data_ptr(), x.data_ptr(), g_recon.data_ptr(),
                                         dz.data_ptr(), n, logits.numel() // n, _stream()),
                   "pg_bce_logits_bwd")
        dkl = torch.empty(n, device=logits.device, dtype=torch.float32)
        g_kl = _chk(g_kl.reshape(1), "elbo.gk")
        _lib.check(lib.pg_fill_scaled(g_kl.data_ptr(), 1.0 / n, dkl.data_ptr(), n, _stream()),
                   "pg_fill_scaled")
        return dz, None, dkl


def elbo_terms(logits, x, kl):
    """Returns (recon_loss.mean(), kl_div.mean()) of the reference VAE loss_fn (vae.py:149-159)."""
    return _ElboMean.apply(logits, x, kl)


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

