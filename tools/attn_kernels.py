"""Kernel-only timing of the three causal-attention launches (C-ABI + HIP events), ImageGPT shape.
usage: python tools/attn_kernels.py [batch] [iters]     (the kernel-family switches — PG_ATTN_MFMA=0 for the VALU kernels, ... — exist only in the `ab` library:
PG_VARIANT=ab python pytorch-generative_amd/build.py, then PG_HIP_LIB=<lib/libpg_hip_ab.so> PG_ATTN_MFMA=0 python tools/attn_kernels.py;
the production library compiles them out and this tool refuses to run with one of them set against it)"""


def _refuse_dead_switches():
    """A PG_* kernel switch set against the PRODUCTION library would silently measure the default kernel (csrc/common.h: PG_AB_ENV is a
    compile-time null there)."""
    import os

    dead = [k for k in os.environ if k.startswith(("PG_ATTN_", "PG_CONV_", "PG_WGRAD_")) and k != "PG_CONV_LOG"]
    lib = os.environ.get("PG_HIP_LIB", "")
    if dead and "_ab" not in os.path.basename(lib):
        raise SystemExit(f"{dead} set, but the library in use is not an ab build (PG_HIP_LIB={lib or 'unset'}): these switches are "
                         "compiled out of the production library — build one with PG_VARIANT=ab python pytorch-generative_amd/build.py")


_refuse_dead_switches()
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-generative_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
r = bench.attention_kernel_roofline(batch, torch.device("cuda:0"), iters=iters)
print(json.dumps({k: {"ms": round(v["launch_ms"], 4), "tflops": round(v["tflops"], 2),
                      "frac": round(v["tflops"] / bench.FP32_PEAK_TFLOPS, 4)} for k, v in r.items()}))
