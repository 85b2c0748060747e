import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libcoexec.so"))
lib.coexec_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
names = ["A=MFMA alone", "B=VALU alone", "A=MFMA | B=VALU", "both: 24 MFMA then 48 VALU", "same, role B rotated", "A=MFMA prio 3 | B=VALU",
         "both, setprio around MFMA", "both, role B prio 1", "both, role B prio 3", "both, prio alternates per iteration"]
iters = 4000
st = torch.cuda.current_stream().cuda_stream
for mode, name in enumerate(names):
    out = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    lib.coexec_run(out.data_ptr(), 50, mode, 256, st); torch.cuda.synchronize()
    out.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.coexec_run(out.data_ptr(), iters, mode, 256, st); e1.record(); torch.cuda.synchronize()
    o = out.view(256, 8).double()
    a = o[:, :4][o[:, :4] > 0]; b = o[:, 4:][o[:, 4:] > 0]
    fa = f"{a.mean() / iters:8.1f}" if a.numel() else "      - "
    fb = f"{b.mean() / iters:8.1f}" if b.numel() else "      - "
    print(f"mode {mode} {name:32s} {e0.elapsed_time(e1) * 1e3:9.1f} us   cycles per iteration: role A {fa}  role B {fb}"
          f"   (24 MFMA = 384 cycles at 16/instr; 48 VALU = 192 at 4/instr)")
