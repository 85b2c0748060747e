"""ops.attention — causal attention (separate q / kv and merged qkv), merged projection weights, the determinism switch.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import _chk, _sink, _stream, zeros


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
