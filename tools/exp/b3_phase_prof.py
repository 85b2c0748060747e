"""Per-phase cycle counts of conv_b3_kernel from its own clocks (ablation build: PG_ABLATE=1 python
pytorch-generative_amd/build.py -> lib/libpg_hip_ablate.so, conv_b3_kernels.h PG_PROF_*): for every wave the cycles
spent in {MFMA loop, barrier after it, commit (activation + split + LDS writes), load issue, epilogue, barrier
before the next MFMA loop}, summed over its steps.   usage: python tools/exp/b3_phase_prof.py [N ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

from pytorch_generative_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpg_hip_ablate.so")
from pytorch_generative_amd import ops  # noqa: E402

lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.pg_b3_set_prof.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
NAMES = ["mfma", "bar1", "commit", "issue", "epilogue", "bar2", "total", "steps"]


def one(batch, cin, cout, hw, k, act):
    """k: an int (k x k window: 3 -> pad 1, else pad 1 cropped) or (kh, kw, pad_h, pad_w); output cropped to hw x hw."""
    if isinstance(k, int):
        k = (k, k, k // 2 if k == 3 else (1 if k > 1 else 0), k // 2 if k == 3 else (1 if k > 1 else 0))
    spec = ops.ConvSpec(*k)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, cin, hw, hw, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k[0], k[1], generator=g) * 0.05).to(dev)
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(batch, cout, hw, hw, device=dev)
    fmt = ops._use_mfma(lib, cin, cout, spec, (hw, hw), hw)
    wfrag = ops._pack_frag(lib, wt, spec, False, fmt)
    T = len(spec.fwd_taps)
    prof = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()

    def run():
        _lib.check(lib.pg_conv2d_mfma(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), 0, out.data_ptr(), batch, cin,
                                      hw, hw, cout, hw, hw, T, spec.f_dr, spec.f_dc, act, 0, ops.ACT_NONE,
                                      ops.ACT_NONE, fmt, st.cuda_stream), "pg_conv2d_mfma")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    raw.pg_b3_set_prof(prof.data_ptr())
    run()
    torch.cuda.synchronize()
    raw.pg_b3_set_prof(None)
    p = prof.view(-1, 8).cpu()
    p = p[p[:, 6] > 0].double()
    mean = p.mean(0)
    print(f"N={batch} {cin}->{cout} {k[0]}x{k[1]} {hw}x{hw} act={act}: {us:.1f} us/launch (no clocks), {p.shape[0]} waves, "
          f"{int(mean[7])} steps/wave, wave lifetime {mean[6]:.0f} cycles (min {p[:, 6].min():.0f} max {p[:, 6].max():.0f})")
    for i in range(6):
        print(f"    {NAMES[i]:9s} {mean[i]:9.0f} cycles = {100 * mean[i] / mean[6]:5.1f} %  per step {mean[i] / mean[7]:7.0f}"
              f"   (min {p[:, i].min():.0f} max {p[:, i].max():.0f})")
    rest = mean[6] - mean[:6].sum()
    print(f"    prologue+rest {rest:9.0f} cycles = {100 * rest / mean[6]:5.1f} %")
    if os.environ.get("PG_PROF_BY_WAVE"):  # conv_b3q_kernel (16 waves per workgroup): the mean per wave index of a workgroup
        W = int(os.environ["PG_PROF_BY_WAVE"])
        pw = prof.view(-1, 8).cpu().double()
        pw = pw[: (pw.shape[0] // W) * W].view(-1, W, 8)
        live = pw[:, 0, 6] > 0
        pw = pw[live].mean(0)
        for w in range(W):
            print(f"      wave {w:2d}: " + "  ".join(f"{NAMES[i]} {pw[w, i] / max(pw[w, 7], 1):6.0f}" for i in range(6)))


if __name__ == "__main__" and sys.argv[1:2] == ["q"]:
    # conv_b3q_kernel (round 6; its phases: mfma = MFMA block, bar1 = barrier, commit = side work, issue = load retire + DMA issue,
    # epilogue, bar2 = wait for the slab)
    for cin, cout, k in ((256, 256, 1), (128, 256, (1, 2, 0, 1)), (256, 256, (2, 1, 1, 0)), (128, 256, (1, 3, 0, 1))):
        one(512, cin, cout, 32, k, ops.ACT_NONE)
    one(1024, 64, 128, 32, (2, 2, 1, 1), ops.ACT_ELU)
    one(64, 160, 320, 32, (2, 3, 1, 1), ops.ACT_NONE)
    sys.exit(0)
if __name__ == "__main__" and sys.argv[1:2] == ["gated"]:
    # the shapes of GatedPixelCNN's layers (wide kernel, batch 512) and PixelCNN++'s (160 filters, batch 64)
    for cin, cout, k in ((256, 256, 1), (128, 256, 1), (128, 128, 1), (128, 256, (1, 2, 0, 1)), (128, 256, (2, 1, 1, 0)),
                         (128, 128, (1, 3, 0, 1))):
        one(512, cin, cout, 32, k, ops.ACT_NONE)
    one(64, 320, 160, 32, (2, 3, 1, 1), ops.ACT_NONE)
    one(64, 320, 320, 32, (2, 2, 1, 1), ops.ACT_NONE)
    one(128, 32, 32, 64, 3, ops.ACT_NONE)
    sys.exit(0)
if __name__ == "__main__":
    batches = [int(v) for v in sys.argv[1:]] or [512]
    for b in batches:
        one(b, 64, 64, 32, 2, ops.ACT_ELU)     # PixelSNAIL ResidualBlock 2x2 64 -> 64 (conv_b3_kernel<4, 4>)
        one(b, 64, 64, 32, 2, ops.ACT_NONE)
    one(batches[0], 64, 128, 32, 2, ops.ACT_ELU)  # the wide (CG = 2) kernel
    one(batches[0] // 4, 32, 32, 64, 3, ops.ACT_NONE)  # VD-VAE / beta-VAE 3x3 32 -> 32 on 64x64 (conv_b3_kernel<2, NT>)
