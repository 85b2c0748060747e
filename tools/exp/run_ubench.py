import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libubench.so"))
vp = ctypes.c_void_p
lib.ub_run.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
dev = torch.device("cuda:0")
out = torch.empty(256 * 8 * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = ["mfma4x4x1", "mfma16x16x4", "v_exp", "v_fma", "mfma4+exp", "mfma16+exp", "mfma4+fma", "v_rcp", "pk_fma", "pk_fma+mfma16", "pk_add", "mfma16+2fma", "bf16_16x16x32", "bf16mfma+exp", "bf16mfma+2fma", "bf16mfma+mfma4", "bf16mfma+4exp", "bf16 chain x1", "bf16 chain x2", "bf16 chain x4"]
iters = 20000
for threads in (256, 512, 1024):
    wps = threads // 256
    for mode, name in enumerate(names):
        if mode < 12 and len(sys.argv) > 1: continue
        lib.ub_run(mode, out.data_ptr(), 100, 256, threads, st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.ub_run(mode, out.data_ptr(), iters, 256, threads, st); b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        # per SIMD: wps waves each issuing iters*8 "units"
        cyc = ms * 1e-3 * 2.4e9 / (iters * 8 * wps)
        print(f"waves/SIMD={wps} {name:12s} {ms:8.3f} ms  {cyc:6.2f} cyc per unit per SIMD (at 2.4 GHz)")
