"""Kernel-only timing of the three causal-attention launches (C-ABI + HIP events), ImageGPT shape.
usage: python tools/attn_kernels.py [batch] [iters]     (PG_ATTN_MFMA=0 selects the VALU kernels)"""
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
