"""Two processes on ONE GPU, every workload bench.py times, forward AND backward: results stay bit-identical from repeat to repeat.

Round 5: `conv_mfma_kernel` zero-filled its LDS tiles and committed the first staged chunk without a barrier in between; alone on the
GPU the fill always won, next to a twin process a late wave zeroed entries that were already committed (2-4 of 15 forwards wrong,
profiles/README.md round 5 item 16) — and 275 green single-process tests never saw it. The kernels of the bit-reproducible set are
deterministic, so ANY difference between two runs of the same step on the same batch with the same parameters is a race; this tier runs
all eight bench.WORKLOADS (reference models: pixel_cnn.py, gated_pixel_cnn.py, pixel_snail.py, image_gpt.py, vae.py, vd_vae.py) as twins,
each process comparing every module output and its whole flat gradient buffer with its own first repeat (tests/twin_worker.py)."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPEATS = 8
# two pairs of processes (import + context start-up is paid once per pair); within a pair the twins meet before each workload
GROUPS = [["pixel_snail", "image_gpt", "pixel_cnn", "beta_vae"],
          ["gated_pixel_cnn", "vd_vae", "pixel_cnn_pp", "image_gpt_repro"]]


@pytest.mark.gpu
@pytest.mark.parametrize("group", GROUPS, ids=["+".join(g) for g in GROUPS])
def test_twin_processes_do_not_disturb_each_other(tmp_path, group):
    me = os.path.join(HERE, "twin_worker.py")
    procs = [subprocess.Popen([sys.executable, me, f"twin{i}", str(tmp_path), "2", str(REPEATS), *group],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(2)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    for i, out in enumerate(outs):
        verdicts = dict((m[0], (int(m[1]), int(m[2]), int(m[3]))) for m in
                        re.findall(rf"^\[twin{i}\] (\w+): (\d+) of (\d+) repeats differ \((\d+) tensors", out, flags=re.M))
        assert sorted(verdicts) == sorted(group), out[-3000:]
        for name, (bad, n, tensors) in verdicts.items():
            assert n == REPEATS - 1 and tensors > 3, (name, n, tensors)
            assert bad == 0, f"{name} (twin {i}): {bad} of {n} repeats differ from the first one\n" + out[-3000:]
