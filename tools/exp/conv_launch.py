"""A few launches of PixelSNAIL's dominant convolution (2x2 64 -> 64, ELU prologue, forward) at the bench's batch, for
rocprofv3 passes: python tools/exp/conv_launch.py [batch] [launches]. Also launches the calibration kernel (ops.add on
the same tensor size: reads 2 x, writes 1 x the activation bytes) so that the PMC units can be checked in the same run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from pytorch_generative_amd import ops  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
a = torch.randn(batch, 64, 32, 32, device=dev)
b = torch.randn(batch, 64, 32, 32, device=dev)
for _ in range(launches):
    ops.add(a, b)
torch.cuda.synchronize()
orig = bench._event_time
bench._event_time = lambda fn, stream, iters=10: orig(fn, stream, iters=launches)
r = bench.conv_kernel_roofline(batch, dev)
print({"batch": batch, **r})
