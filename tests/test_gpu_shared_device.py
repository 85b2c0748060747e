"""Two processes on ONE GPU: every module output of PixelSNAIL's forward stays bit-identical from repeat to repeat.

Round 5: `conv_mfma_kernel` zero-filled its LDS tiles and committed the first staged chunk without a barrier in between; alone on the
GPU the fill always won, next to a twin process a late wave zeroed entries that were already committed (2-4 of 15 forwards wrong,
profiles/README.md round 5 item 16). This is the check that found it: the kernels are deterministic, so ANY difference between two
forwards of the same batch with the same parameters is a race."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_twin_processes_do_not_disturb_each_other():
    env = dict(os.environ, MODE="twins")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", "conc_forward_selfcheck.py"), "pixel_snail", "12"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    verdicts = re.findall(r"^\[(\w+)\] (\d+) of (\d+) repeats differ", out.stdout, flags=re.M)
    assert sorted(v[0] for v in verdicts) == ["alone", "shared0", "shared1"], out.stdout[-2000:]
    assert all(int(bad) == 0 and int(n) == 11 for _, bad, n in verdicts), out.stdout[-3000:]
