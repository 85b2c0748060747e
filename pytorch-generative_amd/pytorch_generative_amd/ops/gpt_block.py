"""ops.gpt_block — the fused position-wise MLP, the ImageGPT block's head / tail kernels, NCHW LayerNorm.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import os

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import _chk, _p, _sink, _stream, zeros, zeros_like


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


_open_chains = []  # weak references to the queues handed out by new_block_chain(): optim.FlatAdam.step() checks that none is pending


def new_block_chain():
    """A queue for the deferred weight-gradient reductions of a model's blocks (see gpt_block_head)."""
    import weakref

    class _Chain(dict):
        pass

    chain = _Chain(jobs=[])
    _open_chains[:] = [r for r in _open_chains if r() is not None]
    _open_chains.append(weakref.ref(chain))
    return chain


def assert_no_pending_block_reductions():
    """Raises if a backward pass left deferred reductions unflushed (the block that flushes never ran backward): the queued
    gradients would silently be missing from the step."""
    for r in _open_chains:
        c = r()
        if c is not None and c["jobs"]:
            raise RuntimeError(f"{len(c['jobs'])} deferred GPT-block weight-gradient reductions were never flushed: the first "
                               "block's backward did not run (set PG_BLOCK_REDUCE_MERGED=0 to reduce per block)")


def flush_block_reductions(chain, n, c, L):
    """Adds the partial weight-gradient rows of every block queued in `chain` (8 blocks per launch) and empties the queue."""
    import ctypes

    lib = _lib.load()
    jobs = chain["jobs"]
    for i in range(0, len(jobs), 8):
        grp = jobs[i:i + 8]
        hw = (ctypes.c_void_p * len(grp))(*[j[0].data_ptr() for j in grp])
        tw = (ctypes.c_void_p * len(grp))(*[j[1].data_ptr() for j in grp])
        gr = (ctypes.c_void_p * (14 * len(grp)))(*[ptr for j in grp for ptr in j[2]])
        _lib.check(lib.pg_gpt_blocks_reduce(len(grp), hw, tw, gr, n, c, L, _stream()), "pg_gpt_blocks_reduce")
    jobs.clear()


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
        chain = ctx.pair.get("chain") if ctx.pair is not None else None
        if pending is not None and chain is not None and all(r is None for r in ret):
            # round 6: a model-level chain of blocks (ImageGPT passes one list to all of its blocks): this block's head kernel
            # leaves its partial rows as well, and the LAST block to run backward (the model's first) adds the rows of every
            # block with ONE launch (pg_gpt_blocks_reduce) instead of one reduce launch per block
            t_ws, t = pending
            _lib.check(lib.pg_gpt_block_head_bwd_partial(*head_args[:8], *head_args[14:], _stream()),
                       "pg_gpt_block_head_bwd_partial")
            # 14 destinations in the C-ABI's order: head lnw, lnb, wq, bq, wkv, bkv | tail w1, b1, w2, b2, wp, bp, lnw, lnb
            chain["jobs"].append((ws, t_ws, [g.data_ptr() for g in tgt] + [t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(),
                                                                          t[7].data_ptr(), t[0].data_ptr(), t[1].data_ptr(),
                                                                          t[2].data_ptr(), t[3].data_ptr()], (tgt, t)))
            if ctx.pair.get("flush"):
                flush_block_reductions(chain, n, c, h * w)
            return (dx, *ret, None, None, None)
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
    kernels share one weight-gradient reduction launch. pair["chain"] = {"jobs": []} shared by ALL blocks of a model and
    pair["flush"] = True on the block whose backward runs last (the model's first block): one reduction launch per 8 blocks."""
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
