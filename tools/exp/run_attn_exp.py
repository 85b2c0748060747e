import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libattn_exp.so")
lib = ctypes.CDLL(so)
N, heads, L = 512, 4, 784
dev = torch.device("cuda:0")
q = torch.randn(N, 16, L, device=dev); k = torch.randn(N, 16, L, device=dev); v = torch.randn(N, 16, L, device=dev)
o = torch.empty_like(q)
vp = ctypes.c_void_p
lib.exp_attn_fwd.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
names = ["v1 CH8", "v1 CH4", "v1 CH4 prefetch", "v2 CH4", "v2 CH4 prefetch", "v2 CH8", "v1 CH2 prefetch", "v3 LDS qpl2", "v3 LDS qpl1", "v3 LDS qpl4", "v4 MFMA QT8", "v4 MFMA QT4", "v5 balanced pairs"]
st = torch.cuda.current_stream().cuda_stream
ref = None
for var, name in enumerate(names):
    for _ in range(2):
        rc = lib.exp_attn_fwd(var, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), N, heads, L, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.exp_attn_fwd(var, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), N, heads, L, st)
    b.record(); torch.cuda.synchronize()
    chk = float(o.double().sum())
    print(f"{name:18s} rc={rc} {a.elapsed_time(b)/5:.3f} ms  checksum {chk:.4f}")
