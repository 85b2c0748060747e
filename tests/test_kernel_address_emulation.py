"""CPU restatements of the global address generation of three bf16x3 kernels (tools/exp/emu_wgrad_bounds.py,
tools/exp/emu_pw_bounds.py): every emulated read / write stays inside its tensor for the shapes the models launch at
batch 1 and 2 (the index logic itself, not the kernels, is what runs here)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "exp", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("kernel", [(3, 3, 1, 1), (2, 2, 1, 1), (1, 3, 0, 1), (2, 1, 2, 0), (1, 2, 0, 1), (1, 1, 0, 0)])
def test_weight_gradient_kernels_read_in_bounds(kernel):
    emu = _load("emu_wgrad_bounds")
    taps = emu.taps_of(*kernel)
    took = 0
    for n in (1, 2):
        for (h, w) in ((28, 28), (12, 12), (10, 20), (16, 16), (8, 64)):
            # (64, 64) / (64, 128) / (192, 128): the round-5 big tiles (64 x channels per workgroup; 128 dy channels for one tap)
            for (cin, cout) in ((32, 32), (32, 64), (128, 256), (64, 64), (64, 128), (192, 128)):
                took += emu.old_kernel(n, cin, cout, h, w, taps) is not None
                took += emu.b3s_kernel(n, cin, cout, h, w, taps) is not None
    assert took > 0


def test_pointwise_kernel_addresses_in_bounds():
    emu = _load("emu_pw_bounds")
    took = 0
    for n in (1, 2):
        for (h, w) in ((28, 28), (16, 16), (18, 16)):
            for cin in (16, 24, 32, 64):
                for cout in (16, 36, 64, 72, 96, 128):
                    took += emu.pw(n, cin, h, w, cout) is not None
    assert took > 0
