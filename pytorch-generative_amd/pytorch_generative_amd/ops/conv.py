"""ops.conv — convolution over a tap list (forward, data / weight gradient, fused epilogues, pass-through aliases).

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import os

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import (
    ACT_ELU,
    ACT_ELU_OUT,
    ACT_NONE,
    ACT_RELU,
    CONV_FMT_B3,
    CONV_FMT_B3_GATE,
    FUSE_SKIP,
    _chk,
    _dense_per_image,
    _p,
    _sink,
    _stream,
    zeros,
    zeros_like,
)
from pytorch_generative_amd.ops.elementwise import add


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


class GradSlot:
    """One gradient tensor handed from a consumer's backward to a producer's backward outside autograd's own edges: the "dual"
    data gradient (conv2d_taps(..., in_sum=(r, slot))) computes TWO gradients for its one input x = elu(a) + r — autograd carries
    the first (a's), the second (r's, already multiplied by r's ELU derivative) waits here for the backward of the convolution that
    added r (conv2d_taps(..., out_pre_scaled=True, res=r, res_slot=slot)), which returns it as r's gradient. Filled and emptied
    once per backward pass; taking from an empty slot raises (the consumer's backward did not run first, or took another path)."""

    def __init__(self):
        self._g = None

    def put(self, g):
        if self._g is not None:
            raise RuntimeError("GradSlot: filled twice in one backward pass")
        self._g = g

    def take(self):
        g, self._g = self._g, None
        if g is None:
            raise RuntimeError("GradSlot: empty — the dual data gradient that fills it has not run")
        return g


class _ConvTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, res, spec, out_hw, in_act, gw, gb, out_act=ACT_NONE,
                out_pre_scaled=False, in_post=ACT_NONE, n_skip=0, res2=None, gate=None, gate_res=None, res_slot=None,
                in_sum=None):
        lib = _lib.load()
        ctx.gate = gate
        ctx.res_slot, ctx.in_sum_slot = res_slot, None
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
        if out_pre_scaled and res is not None and res_slot is not None:
            # the consumer is a dual data gradient (in_sum): it recovers this activation's output as (its input - res) and hands
            # res's gradient back through the slot
            if out_act != ACT_ELU or res2 is not None:
                raise ValueError("conv2d: res_slot needs out_act='elu' and a single residual")
        elif out_pre_scaled and res is not None:
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
        if gate is not None:
            # GatedActivation (+ the residual behind it) in the convolution's epilogue: `out` keeps the pre-gate values for
            # backward, y is what the caller sees
            if mfma != CONV_FMT_B3 or res2 is not None or out_act != ACT_NONE:
                raise ValueError("conv2d: gate= needs a bf16x3 convolution with at most one residual (check ops.conv_gate_ok first)")
            if gate_res is not None:
                gate_res = _chk(gate_res, "conv2d.gate_res")
                if tuple(gate_res.shape) != (n, cout // 2, oh, ow):
                    raise ValueError("conv2d: gate_res shape mismatch")
            fmt_t = _use_mfma(lib, cout, cin, spec, (ih, iw), ow) if ctx.needs_input_grad[0] else 0
            # forward fragments in the gate-interleaved channel order (each wave owns both halves of its gate channels)
            if fmt_t:
                wfrag, ctx.wfrag_t = _pack_frag_both(lib, weight, spec, CONV_FMT_B3_GATE, fmt_t)
            else:
                wfrag = _pack_frag(lib, weight, spec, False, CONV_FMT_B3_GATE)
            y = torch.empty((n, cout // 2, oh, ow), device=x.device, dtype=torch.float32)
            _lib.check(
                lib.pg_conv2d_mfma_gate(
                    x.data_ptr(), wfrag.data_ptr(), _p(bias), _p(res), out.data_ptr(), n, cin, ih, iw, cout, oh, ow,
                    len(spec.fwd_taps), spec.f_dr, spec.f_dc, in_act, gate, _p(gate_res), y.data_ptr(), _stream(),
                ),
                "pg_conv2d_mfma_gate",
            )
            ctx.save_for_backward(x, weight, out)
            # (the convolution's own residual is part of the stored pre-gate tensor; its gradient is the gate's input gradient)
            ctx.spec, ctx.in_act, ctx.has_bias, ctx.has_res, ctx.has_res2 = spec, in_act, bias is not None, res is not None, False
            ctx.has_gate_res = gate_res is not None
            ctx.gw, ctx.gb = gw, gb
            ctx.out_act, ctx.out_pre_scaled, ctx.in_post = ACT_NONE, False, in_post
            ctx.n_skip = n_skip
            if n_skip:
                return (y,) + tuple(x.view_as(x) for _ in range(n_skip))
            return y
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
        if in_sum is not None:
            r_sum, ctx.in_sum_slot = in_sum
            if gate is not None or n_skip or in_act != ACT_ELU or tuple(r_sum.shape) != tuple(x.shape):
                raise ValueError("conv2d: in_sum needs in_act='elu', no skip aliases and r of the input's shape")
            if not conv_dual_ok(x, weight, spec):
                raise ValueError("conv2d: in_sum on a shape the dual data gradient does not take (check ops.conv_dual_ok first)")
            r_sum = _chk(r_sum, "conv2d.in_sum")
        if out_act != ACT_NONE and not out_pre_scaled:
            # backward recovers act' from the output: v = out - res
            tensors = (x, weight, out, res) if res is not None else (x, weight, out)
        else:
            tensors = (x, weight)
        ctx.n_own = len(tensors)
        ctx.save_for_backward(*(tensors + ((r_sum,) if in_sum is not None else ())))
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
        d_gate_res = None
        if ctx.gate is not None:
            # through the gate first: dz (both halves) from the stored pre-gate output, then the usual data / weight gradients
            lib = _lib.load()
            z = ctx.saved_tensors[2]
            dy = _chk(dy, "conv2d.dy")
            nz, c2, hz, wz = z.shape
            dz = torch.empty_like(z)
            _lib.check(lib.pg_gated_bwd(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), nz, c2 // 2, hz * wz, ctx.gate,
                                        _stream()), "pg_gated_bwd")
            d_gate_res = dy if ctx.has_gate_res else None
            dy = dz
        return _ConvTaps.backward_impl(ctx, dy, need[0], need[1], ctx.has_bias and need[2],
                                       d_skips=d_skips) + (None, None, None, None,
                                                           dy if getattr(ctx, "has_res2", False) else None,
                                                           None, d_gate_res, None, None)

    @staticmethod
    def backward_impl(ctx, dy, need_dx, need_w, need_b, d_skips=()):
        lib = _lib.load()
        x, weight = ctx.saved_tensors[:2]
        spec = ctx.spec
        dy = _chk(dy, "conv2d.dy")
        dres = dy if ctx.has_res else None
        if ctx.has_res and getattr(ctx, "res_slot", None) is not None:
            dres = ctx.res_slot.take()  # the residual's gradient, already through ITS producer's ELU (dual data gradient)
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
            slot = getattr(ctx, "in_sum_slot", None)
            if slot is not None:
                # dual: dx = the gradient of a's pre-activation (x = elu(a) + r), the second output that of r's producer
                r_sum = ctx.saved_tensors[ctx.n_own]
                dx2 = torch.empty_like(x)
                _lib.check(lib.pg_conv2d_mfma_dual(dy.data_ptr(), wfrag_t.data_ptr(), dx.data_ptr(), dx2.data_ptr(), n, cout, oh, ow,
                                                   cin, x.data_ptr(), dact, r_sum.data_ptr(), _stream()), "pg_conv2d_mfma_dual")
                slot.put(dx2)
            else:
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
                in_post=ACT_NONE, n_skip=0, res2=None, gate=None, gate_res=None, res_slot=None, in_sum=None):
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
                            _sink(bias_param), out_act, bool(out_pre_scaled), in_post, 0, res2, gate, gate_res, res_slot,
                            in_sum)
        return (y,) + (x,) * int(n_skip)
    return _ConvTaps.apply(x, weight, bias, res, spec, tuple(out_hw), in_act,
                           _sink(weight_param), _sink(bias_param), out_act, bool(out_pre_scaled),
                           in_post, int(n_skip), res2, gate, gate_res, res_slot, in_sum)


FUSE_DUAL = os.environ.get("PG_FUSE_DUAL", "1") != "0"  # A/B: 0 = pg_act_bwd_from_out launches behind the block tail's convolutions


def conv_dual_ok(x, weight, spec):
    """True if conv2d_taps(x, ..., in_act=elu, in_sum=(r, slot)) can deliver both producers' gradients from its data
    gradient's epilogue (pg_conv_dual_ok: a 1x1 convolution whose data gradient runs on the bf16x3 1x1 kernel)."""
    if not FUSE_DUAL or not x.is_cuda or x.dtype != torch.float32 or tuple(weight.shape[2:]) != (1, 1):
        return False
    if len(spec.fwd_taps) != 1 or spec.f_dr[0] != 0 or spec.f_dc[0] != 0:
        return False
    lib = _lib.load()
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    h, w = int(x.shape[2]), int(x.shape[3])
    if _use_mfma(lib, cout, cin, spec, (h, w), w) != CONV_FMT_B3:
        return False
    return bool(lib.pg_conv_dual_ok(cout, cin, h, w))


FUSE_GATE = os.environ.get("PG_FUSE_GATE", "1") != "0"  # A/B: 0 = the standalone gate kernel behind the convolution


def conv_gate_ok(x, weight, spec, out_hw=None):
    """True if conv2d_taps(..., gate=...) can fuse the GatedActivation that follows this convolution into its launch
    (pg_conv_gate_fusable: bf16x3 format on the wide kernel, a multiple of 128 output channels)."""
    if not FUSE_GATE or not x.is_cuda or x.dtype != torch.float32:
        return False
    lib = _lib.load()
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    oh, ow = out_hw if out_hw is not None else spec.full_out(x.shape[2], x.shape[3])
    if _use_mfma(lib, cin, cout, spec, (oh, ow), x.shape[3]) != CONV_FMT_B3:
        return False
    return bool(lib.pg_conv_gate_fusable(cin, cout, oh, ow, len(spec.fwd_taps), spec.f_dr, spec.f_dc))
