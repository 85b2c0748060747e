"""Flat-buffer Adam: the optimiser half of the reference's timed step on one HIP launch chain.

The reference step (trainer.py:183-191) is `clip_grad_norm_` over ~100-600 small tensors, then
`torch.optim.Adam.step()`. Here all parameters live in ONE contiguous fp32 buffer (each
nn.Parameter becomes a view into it), all gradients in a second one (`param._pg_grad` /
`param.grad` are views; the backward kernels accumulate straight into it and RCCL all-reduces
it in one message), and the whole norm + clip + Adam + lr-decay sequence is 4 kernels driven by
a device-side state block — capturable in a hipGraph, no host sync.
"""

import torch

from pytorch_generative_amd import _lib

# layout of the device state block (see include/pg_hip.h)
_STEP, _LR, _SUMSQ, _NORM, _COEF, _LRMUL, _MAXNORM, _PRESCALE = range(8)


# the keys torch.optim.Adam expects in a param_group besides lr / betas / eps (none is implemented
# here beyond its default: load_state_dict rejects anything else)
_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                            differentiable=False, fused=None, decoupled_weight_decay=False)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def plan_layout(params):
    """Offsets (in floats) of every parameter in the flat buffers, and the total length.

    Every slice starts 16-byte aligned so float4 paths apply to each view — except that a parameter
    which declares `_pg_follows = other` is placed directly behind `other` (no padding; `other` must
    have a multiple of 4 elements): modules use this to get zero-copy merged views of weights they
    want to run as one convolution (nn.CausalAttention: q and kv projections)."""
    ids = {id(p) for p in params}
    followers = {}
    for p in params:
        lead = getattr(p, "_pg_follows", None)
        if lead is not None and id(lead) in ids and lead.numel() % 4 == 0 and id(lead) not in followers:
            followers[id(lead)] = p
    placed, off_of, total = set(), {}, 0
    for p in params:
        if id(p) in placed:
            continue
        lead = getattr(p, "_pg_follows", None)
        if lead is not None and followers.get(id(lead)) is p and id(lead) not in placed:
            continue  # placed right after its leader below
        q = p
        while q is not None and id(q) not in placed:
            off_of[id(q)] = total
            placed.add(id(q))
            nxt = followers.get(id(q))
            total += q.numel() if nxt is not None else (q.numel() + 3) // 4 * 4
            q = nxt
    return [off_of[id(p)] for p in params], total


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, bias correction; no weight decay / amsgrad) over a
    flattened parameter set, with the global grad-norm (clip_grad_norm_) fused in front.

    Args:
        params: iterable of parameters (a single param group).
        lr, betas, eps: as torch.optim.Adam.
        max_norm: clip threshold for the global L2 grad norm (None = the reference's 1e50,
            i.e. compute/log the norm but never scale).
        lr_decay: per-step multiplicative lr factor applied on the device
            (MultiplicativeLR(lambda _: c) of the reference's reproduce() functions).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None, lr_decay=1.0):
        params = [p for p in params]
        if not params:
            raise ValueError("FlatAdam: empty parameter list")
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam supports a single parameter group")
        self._params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        dev = self._params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: parameters must live on the MI355X (move the model first)")
        for p in self._params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatAdam: all parameters must be float32 on one device")
        self._lib = _lib.load()
        offs, total = plan_layout(self._params)
        self._offsets, self._numel = offs, total
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self._params, offs):
                n = p.numel()
                self.flat_param[o : o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o : o + n].view(p.shape)
                g = self.flat_grad[o : o + n].view(p.shape)
                p._pg_grad = g
                p.grad = g
        st = torch.zeros(8, dtype=torch.float32)
        st[_LR], st[_LRMUL] = lr, lr_decay
        st[_MAXNORM] = float("inf") if max_norm is None else max_norm  # 1e50 overflows fp32: never clips
        st[_PRESCALE] = 1.0
        self.state_block = st.to(dev)
        self._host_lr = lr
        self._lr_decay = float(lr_decay)

    # ---- helpers -------------------------------------------------------------------------
    def set_grad_prescale(self, s):
        """Gradients are multiplied by `s` before the norm/update (1/world after an all-reduce sum)."""
        self.state_block[_PRESCALE] = float(s)

    def set_max_norm(self, max_norm):
        """Clip threshold of the fused global grad-norm (None: compute the norm, never scale)."""
        self.state_block[_MAXNORM] = float("inf") if max_norm is None else float(max_norm)

    def sync_lr_from_groups(self):
        """Pushes param_groups[0]['lr'] (as set by a torch lr_scheduler or by hand) to the device
        state block when it changed. Host-side compare; called by step() in eager mode and by the
        Trainer after lr_scheduler.step() (a hipGraph replay does not run Python)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._host_lr:
            if self._lr_decay != 1.0:
                raise ValueError("FlatAdam: lr_decay (device-side MultiplicativeLR) cannot be combined "
                                 "with a host lr_scheduler that also changes param_groups['lr']")
            self.state_block[_LR] = lr
            self._host_lr = lr

    def current_lr(self):
        """The learning rate the next step will use (read back from the device: with lr_decay the
        host-side param_groups value is only the initial one)."""
        return float(self.state_block[_LR].item())

    def measure_grad_norm(self):
        """Global L2 norm of the (pre-scaled) flat gradient as a device scalar, without stepping."""
        scale = self.state_block[_PRESCALE]
        return self.flat_grad.norm() * scale

    def grad_norm(self):
        """Device scalar: the global L2 norm computed by the last step()."""
        return self.state_block[_NORM]

    def zero_grad(self, set_to_none=False):
        # the library's own fill kernel (no ATen launch inside a captured step)
        _lib.check(_lib.load().pg_fill(self.flat_grad.data_ptr(), 0.0, self.flat_grad.numel(),
                                       torch.cuda.current_stream().cuda_stream), "pg_fill")
        for p in self._params:  # keep the views attached (autograd may have replaced them)
            if p.grad is not p._pg_grad:
                p.grad = p._pg_grad

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise ValueError("FlatAdam does not support closures")
        from pytorch_generative_amd.ops import gpt_block as _gb

        _gb.assert_no_pending_block_reductions()  # gradients parked in a deferred reduction must have reached the flat buffer
        if not all(p.requires_grad for p in self._params):
            raise RuntimeError("FlatAdam: a parameter was frozen after the optimizer was built; the flat "
                               "update covers every slice — rebuild FlatAdam over the trainable set")
        self.sync_lr_from_groups()
        lib, s = self._lib, _stream()
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        _lib.check(lib.pg_sumsq_accum(self.flat_grad.data_ptr(), self._numel,
                                      self.state_block.data_ptr(), s), "pg_sumsq_accum")
        _lib.check(lib.pg_adam_prepare(self.state_block.data_ptr(), s), "pg_adam_prepare")
        _lib.check(
            lib.pg_adam_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                             self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self._numel,
                             self.state_block.data_ptr(), b1, b2, eps, s),
            "pg_adam_step",
        )

    # ---- checkpoint compatibility with torch.optim.Adam's state_dict layout ---------------
    def state_dict(self):
        """torch.optim.Adam's layout (state keyed by the parameter's index in the group, full
        param_group keys) so that the dict loads into torch.optim.Adam and back."""
        step = float(self.state_block[_STEP].item())
        slot = {id(p): (o, p) for p, o in zip(self._params, self._offsets)}
        state = {}
        all_params = self.param_groups[0]["params"]
        for i, p in enumerate(all_params):
            if id(p) not in slot:
                continue
            o, n = slot[id(p)][0], p.numel()
            state[i] = {
                "step": torch.tensor(step),
                "exp_avg": self.exp_avg[o : o + n].view(p.shape).clone(),
                "exp_avg_sq": self.exp_avg_sq[o : o + n].view(p.shape).clone(),
            }
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group.update(_ADAM_GROUP_DEFAULTS)
        group["lr"] = self.current_lr()
        group["params"] = list(range(len(all_params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
            raise ValueError("FlatAdam implements plain Adam: weight_decay / amsgrad / maximize in the "
                             "checkpoint are not supported")
        for k, v in group.items():
            if k != "params":
                self.param_groups[0][k] = v
        self.state_block[_LR] = float(group["lr"])
        self._host_lr = float(group["lr"])
        slot = {id(p): o for p, o in zip(self._params, self._offsets)}
        step = 0.0
        with torch.no_grad():
            for i, p in enumerate(self.param_groups[0]["params"]):
                st = sd["state"].get(i)
                if st is None or id(p) not in slot:
                    continue
                o, n = slot[id(p)], p.numel()
                self.exp_avg[o : o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o : o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, float(st["step"]))
        self.state_block[_STEP] = step
