"""Restatement of one reference training step (trainer.py:173-193) on torch-CPU.
Test infrastructure only: the checker for the HIP step and bench.py's `cpu_baseline` leg."""

import math

import torch

from oracle import models, ops


def is_param(key):
    return not (key.endswith(".mask") or key in ("_c", "_h", "_w"))


def loss_and_grads(forward_fn, state, x, **fwd_kwargs):
    """zero_grad -> forward -> BCE loss -> backward. Returns (logits, loss, grads dict).
    Masks are applied to the weights in place first (nn/convolution.py:42), outside autograd, so
    weight.grad is the unmasked correlation exactly like the reference."""
    models.apply_masks_(state)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in state.items() if is_param(k)}
    p = dict(state)
    p.update(leaves)
    logits = forward_fn(p, x, **fwd_kwargs)
    loss = ops.bce_sum_mean(logits, x)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(leaves.keys(), grads))


def grad_norm(grads):
    """utils.clip_grad_norm_'s total L2 norm over all non-None grads (trainer.py:183-184)."""
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None)
    return math.sqrt(sq)


def adam_step_(state, grads, opt_state, lr, max_norm=1e50, betas=(0.9, 0.999), eps=1e-8):
    """clip_grad_norm_(max_norm) then torch.optim.Adam defaults (trainer.py:183-189), in place.
    opt_state: {"step": int, "m": {k: tensor}, "v": {k: tensor}}; params without a grad are skipped."""
    norm = grad_norm(grads)
    coef = min(1.0, max_norm / (norm + 1e-6))
    opt_state["step"] += 1
    t = opt_state["step"]
    b1, b2 = betas
    bc1, bc2 = 1 - b1**t, 1 - b2**t
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            g = g * coef
            m = opt_state["m"].setdefault(k, torch.zeros_like(g))
            v = opt_state["v"].setdefault(k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            state[k].addcdiv_(m, denom, value=-lr / bc1)
    return norm


def new_opt_state():
    return {"step": 0, "m": {}, "v": {}}
