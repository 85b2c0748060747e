"""Phase ablation of conv_b3q_kernel (ablation library: PG_ABLATE=1 python pytorch-generative_amd/build.py; WRONG results, timing only):
PG_B3_DBG bits  1 no x loads, 2 no commit, 4 no MFMA block, 8 no epilogue, 16 no slab DMA, 32 every wave side-first."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
    import torch
    from pytorch_generative_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpg_hip_ablate.so")
    from pytorch_generative_amd import ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = []
    for batch, cin, cout, k in ((512, 256, 256, (1, 1, 0, 0)), (512, 256, 256, (2, 1, 1, 0)), (1024, 64, 128, (2, 2, 1, 1))):
        spec = ops.ConvSpec(*k)
        x = torch.randn(batch, cin, 32, 32, device=dev)
        wt = torch.randn(cout, cin, k[0], k[1], device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        o = torch.empty(batch, cout, 32, 32, device=dev)
        fmt = ops._use_mfma(lib, cin, cout, spec, (32, 32), 32)
        wfrag = ops._pack_frag(lib, wt, spec, False, fmt)
        T = len(spec.fwd_taps)
        st = torch.cuda.current_stream()
        def run():
            _lib.check(lib.pg_conv2d_mfma(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), 0, o.data_ptr(), batch, cin, 32, 32, cout, 32, 32,
                                          T, spec.f_dr, spec.f_dc, 0, 0, 0, 0, fmt, st.cuda_stream), "conv")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{cin}->{cout} {k[0]}x{k[1]}: {e0.elapsed_time(e1) * 50:7.1f} us")
    print(f"dbg {os.environ.get('PG_B3_DBG', '0'):>3s}: " + "   ".join(out), flush=True)
else:
    for bits in (0, 1, 2, 3, 4, 8, 16, 32, 4 + 8, 1 + 2 + 8 + 16, 1 + 2 + 4 + 8, 1 + 2 + 4 + 8 + 16):
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, PG_B3_DBG=str(bits)), check=True)
