"""ops.phase_weights (the four stride-2 phase kernels of a 4x4 weight, one launch each way: pg_phase_weights / pg_phase_weights_bwd
since round 6 — GPU tests) against the strided slices it replaces (nn/convolution.py: Conv2d._forward_down2,
ConvTranspose2d.forward): values and gradients, with and without a flat-gradient sink; ops.split_in_channels (pg_copy_rows)."""
import pytest
import torch

from pytorch_generative_amd import ops


@pytest.mark.gpu
@pytest.mark.parametrize("transposed", [False, True])
def test_phase_weights_equal_the_slices(transposed):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    w = torch.randn(5, 3, 4, 4, device=dev, requires_grad=True)
    w2 = w.detach().clone().requires_grad_(True)
    outs = ops.phase_weights(w, transposed)
    src = w2.transpose(0, 1) if transposed else w2
    refs = []
    for pr in (0, 1):
        for pc in (0, 1):
            r = src[:, :, (1 - pr)::2, (1 - pc)::2]
            refs.append((r.flip(2, 3) if transposed else r).contiguous())
    coef = [torch.randn_like(r) for r in refs]
    sum((o * c).sum() for o, c in zip(outs, coef)).backward()
    sum((o * c).sum() for o, c in zip(refs, coef)).backward()
    assert all(torch.equal(o, r) for o, r in zip(outs, refs))
    assert torch.equal(w.grad, w2.grad)
    # with a flat-gradient sink (optim.FlatAdam's `_pg_grad` view): the backward ADDS into it and hands autograd no gradient
    w3 = w.detach().clone().requires_grad_(True)
    w3._pg_grad = torch.full_like(w3, 0.5)
    outs3 = ops.phase_weights(w3, transposed)
    sum((o * c).sum() for o, c in zip(outs3, coef)).backward()
    assert w3.grad is None and torch.equal(w3._pg_grad, w2.grad + 0.5)


@pytest.mark.gpu
def test_phase_weights_with_an_unused_phase():
    w = torch.randn(2, 2, 4, 4, device="cuda:0", requires_grad=True)
    outs = ops.phase_weights(w)
    (outs[1].sum() * 2.0).backward()  # three of the four gradients are None
    ref = torch.zeros(2, 2, 4, 4)
    ref[:, :, 1::2, 0::2] = 2.0  # phase (0, 1): rows (1 - 0)::2, cols (1 - 1)::2
    assert torch.equal(w.grad.cpu(), ref)


@pytest.mark.gpu
def test_phase_merge4_equals_the_stacked_merge():
    torch.manual_seed(2)
    dev = torch.device("cuda:0")
    ps = [torch.randn(3, 5, 6, 4, device=dev, requires_grad=True) for _ in range(4)]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    a = ops.phase_merge4(ps)
    b = ops.phase_merge(torch.stack(qs))
    g = torch.randn_like(a)
    a.backward(g)
    b.backward(g)
    assert torch.equal(a, b) and all(torch.equal(p.grad, q.grad) for p, q in zip(ps, qs))
    x = a.detach()
    assert torch.equal(x[:, :, 1::2, 0::2], ps[2].detach())  # phase (pr, pc) = (1, 0)


@pytest.mark.gpu
def test_split_in_channels_equals_the_slices():
    torch.manual_seed(1)
    w = torch.randn(4, 6, 1, 1, device="cuda:0", requires_grad=True)
    w2 = w.detach().clone().requires_grad_(True)
    a, b = ops.split_in_channels(w, 2)
    ca, cb = torch.randn_like(a), torch.randn_like(b)
    ((a * ca).sum() + (b * cb).sum()).backward()
    ((w2[:, :2] * ca).sum() + (w2[:, 2:] * cb).sum()).backward()
    assert torch.equal(a, w2[:, :2]) and torch.equal(b, w2[:, 2:]) and torch.equal(w.grad, w2.grad)
    w3 = torch.randn(4, 6, 1, 1, device="cuda:0", requires_grad=True)
    _, b3 = ops.split_in_channels(w3, 2)
    (b3 * cb).sum().backward()  # the first half never used
    assert torch.equal(w3.grad[:, 2:], cb) and float(w3.grad[:, :2].abs().max()) == 0.0
    # 3x3 weight and a flat-gradient sink: the backward adds into it and hands autograd nothing
    w4 = torch.randn(5, 7, 3, 3, device="cuda:0", requires_grad=True)
    w4._pg_grad = torch.full_like(w4, 0.25)
    a4, b4 = ops.split_in_channels(w4, 3)
    c4a, c4b = torch.randn_like(a4), torch.randn_like(b4)
    ((a4 * c4a).sum() + (b4 * c4b).sum()).backward()
    assert torch.equal(a4, w4[:, :3]) and torch.equal(b4, w4[:, 3:]) and w4.grad is None
    assert torch.equal(w4._pg_grad, torch.cat((c4a, c4b), dim=1) + 0.25)


@pytest.mark.gpu
def test_phase_split4_equals_the_stacked_split():
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    x = torch.randn(2, 3, 8, 12, device=dev, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    ps = ops.phase_split4(x)
    xs = ops.phase_split(x2)
    cs = [torch.randn_like(p) for p in ps]
    sum((p * c).sum() for p, c in zip(ps[:3], cs)).backward()      # the fourth phase unused: its gradient is None
    sum((xs[k] * cs[k]).sum() for k in range(3)).backward()
    assert all(torch.equal(p, xs[k]) for k, p in enumerate(ps)) and torch.equal(x.grad, x2.grad)


@pytest.mark.gpu
def test_fanout_sums_the_readers_gradients_in_one_launch():
    """ops.fanout: k aliases for k readers; the gradient of x is the sum of the readers' gradients, also when one of them is a
    channel slice of a wider tensor (the backward of concat_channels), and when a reader is unused."""
    torch.manual_seed(4)
    dev = torch.device("cuda:0")
    x = torch.randn(3, 4, 6, 8, device=dev, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    other = torch.randn(3, 2, 6, 8, device=dev)
    a, b, c, unused = ops.fanout(x, 4)
    wide = ops.concat_channels([other, b])          # b's gradient comes back as a batch-strided slice
    g_a, g_w, g_c = torch.randn_like(a), torch.randn_like(wide), torch.randn_like(c)
    ((a * g_a).sum() + (wide * g_w).sum() + (ops.relu(c) * g_c).sum()).backward()
    ((x2 * g_a).sum() + (torch.cat([other, x2], 1) * g_w).sum() + (torch.relu(x2) * g_c).sum()).backward()
    assert torch.allclose(x.grad, x2.grad, rtol=0, atol=1e-6)
    # 40 vectors through sum_vectors: two launches with the running sum among the rows
    vs = [torch.randn(7, device=dev, requires_grad=True) for _ in range(40)]
    s = ops.sum_vectors(vs)
    s.backward(torch.ones_like(s))
    assert torch.allclose(s, torch.stack([v.detach() for v in vs]).sum(0), atol=1e-5) and all(float(v.grad.min()) == 1.0 for v in vs)
