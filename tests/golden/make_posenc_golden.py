"""Generates tests/golden/posenc.pt from the REAL reference (run in the build container only):

    python tests/golden/make_posenc_golden.py

`image_positional_encoding` (nn/attention.py:37-57) builds its two coordinate planes with `torch.arange(-0.5, 0.5, 1 / h)` on the
HOST; ATen's vectorised arange rounds a few elements differently from `start + i * step` and differently between vector ISAs. The
fixture pins the bits the reference produces in the build container (AVX-512 ATen, the torch version recorded in the file): the HIP
kernel (`posenc_kernel`, elementwise.hip) must reproduce them exactly — `tests/test_gpu_ops.py::test_positional_encoding_bit_exact`
— whatever host the GPU box has.
"""

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import _ref  # noqa: E402

SHAPES = [(2, 3, 32, 32), (1, 1, 28, 28), (2, 3, 8, 8), (1, 3, 64, 64), (1, 1, 24, 40), (1, 1, 12, 61), (1, 1, 7, 5),
          (1, 1, 128, 96)]


def main():
    ref = _ref.load()
    out = {"torch_version": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(), "cases": {}}
    for shape in SHAPES:
        enc = ref.nn.image_positional_encoding(shape)
        assert enc.shape == (shape[0], 2, shape[2], shape[3]) and enc.dtype == torch.float32
        out["cases"][shape] = enc[:1].clone()  # every image of the batch carries the same two planes
        assert all(torch.equal(enc[i], enc[0]) for i in range(shape[0]))
    path = os.path.join(HERE, "posenc.pt")
    torch.save(out, path)
    print(f"wrote {path}: {len(SHAPES)} shapes, {out['cpu_capability']}, torch {out['torch_version']}")


if __name__ == "__main__":
    main()
