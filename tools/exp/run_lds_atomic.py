import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "liblds_atomic.so"))
vp = ctypes.c_void_p
lib.run.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
out = torch.empty(256 * 1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = ["ds_add_f32 64 addr", "ds_add_f32 16 addr x4", "ds_add_u32 64 addr", "ds_write_b32", "read+write", "ds_add_rtn_f32"]
iters = 4000
for threads in (256, 1024):
    for mode, name in enumerate(names):
        lib.run(mode, out.data_ptr(), 10, 256, threads, st); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.run(mode, out.data_ptr(), iters, 256, threads, st); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        n = iters * 4 * (threads // 64)  # wave-instructions per CU (one workgroup per CU)
        print(f"waves/CU={threads // 64:2d} {name:24s} {ms:8.3f} ms  {ms * 1e-3 * 2.4e9 / n:7.1f} cycles per wave-instruction per CU (at 2.4 GHz)")
