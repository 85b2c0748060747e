"""A/B on the GPU box: fused attention backward (pg_causal_attn_bwd -> attn_bwd_m44_kernel) against the
two-kernel backward (pg_causal_attn_bwd_dq + _dkv), d_k = d_v = 4: values + time per launch.
usage: python tools/exp/attn_bwd_ab.py [batch]   (PG_ATTN_BWD_WAVES / PG_ATTN_BWD_PF select the variant)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
from pytorch_generative_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")


def case(batch, heads, hw, strict, iters=10):
    dk = dv = 4
    e, vd, L = heads * dk, heads * dv, hw * hw
    g = torch.Generator().manual_seed(7)
    mk = lambda c: torch.randn(batch, c, hw, hw, generator=g).to(dev)
    q, kv, d_o = mk(e), mk(e + vd), mk(vd)
    o = torch.empty_like(d_o)
    lse = torch.empty(batch, heads, L, device=dev)
    delta = torch.empty_like(lse)
    st = torch.cuda.current_stream().cuda_stream
    kvs = (e + vd) * L
    _lib.check(lib.pg_causal_attn_fwd(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L, o.data_ptr(),
                                      lse.data_ptr(), batch, heads, L, dk, dv, e * L, kvs, kvs, vd * L,
                                      int(strict), st), "fwd")

    def bwd(fn, dq, dkv):
        _lib.check(fn(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L, o.data_ptr(), d_o.data_ptr(),
                      lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dkv.data_ptr(),
                      dkv.data_ptr() + 4 * e * L, batch, heads, L, dk, dv, e * L, kvs, kvs, vd * L, vd * L,
                      e * L, kvs, kvs, int(strict), st), "bwd")

    dq0, dkv0 = torch.zeros_like(q), torch.zeros_like(kv)
    dq1, dkv1 = torch.full_like(q, 7.0), torch.full_like(kv, 7.0)
    bwd(lib.pg_causal_attn_bwd_dq, dq0, dkv0)
    bwd(lib.pg_causal_attn_bwd_dkv, dq0, dkv0)
    bwd(lib.pg_causal_attn_bwd, dq1, dkv1)
    torch.cuda.synchronize()
    err = lambda a, b: float((a - b).abs().max() / b.abs().max())

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    t_dq = timeit(lambda: bwd(lib.pg_causal_attn_bwd_dq, dq0, dkv0))
    t_dkv = timeit(lambda: bwd(lib.pg_causal_attn_bwd_dkv, dq0, dkv0))
    t_f = timeit(lambda: bwd(lib.pg_causal_attn_bwd, dq1, dkv1))
    print(f"N={batch} heads={heads} L={L} strict={strict}: err dq {err(dq1, dq0):.1e} dk {err(dkv1[:, :e], dkv0[:, :e]):.1e} "
          f"dv {err(dkv1[:, e:], dkv0[:, e:]):.1e} | two-kernel {t_dq:.0f} + {t_dkv:.0f} = {t_dq + t_dkv:.0f} us, "
          f"fused {t_f:.0f} us", flush=True)


big = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
CASES_Q = [(2, 4, 28, 0), (64, 4, 28, 0), (big, 4, 28, 0)]
for args in CASES_Q if os.environ.get("PG_AB_QUICK") else [(2, 4, 8, 0), (3, 4, 8, 1), (2, 4, 28, 0), (2, 4, 28, 1), (2, 4, 32, 0), (5, 2, 12, 1), (64, 4, 28, 0),
             (big, 4, 28, 0)]:
    case(*args)
