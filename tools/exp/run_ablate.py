"""Times the attention launches with the ablated libraries of build_ablate.sh (timing only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-generative_amd"))
import torch
from pytorch_generative_amd import _lib
which = sys.argv[1]
if which != "0":
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", f"libpg_abl{which}.so")
import bench
r = bench.attention_kernel_roofline(1024, torch.device("cuda:0"), iters=10)
print("abl", which, json.dumps({k: round(v["launch_ms"], 4) for k, v in r.items()}))
