"""CPU: host-side logic — the C-ABI library loads and exports every symbol the header declares,
the module surface matches the reference's state_dict layout, argument errors follow the
reference's convention, and the flat-gradient all-reduce is correct under gloo (world_size 2)."""

import os
import re

import pytest
import torch

import _util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    from pytorch_generative_amd import _lib

    header = open(os.path.join(ROOT, "include", "pg_hip.h")).read()
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pg_abi_version() == _lib.ABI_VERSION
    assert lib.pg_conv_b_pad(1) == 16 and lib.pg_conv_b_pad(17) == 32
    assert lib.pg_packed_weight_floats(3, 5, 20) == 3 * 5 * 32


def test_argument_errors_without_gpu(lib):
    """Status codes -> exceptions; no kernel is launched for rejected arguments."""
    from pytorch_generative_amd import _lib

    ia = _lib.int_array
    rc = lib.pg_conv2d_taps(0, 0, 0, 0, 0, 1, 1, 4, 4, 1, 4, 4, 1, ia([0]), ia([0]), 0, 0, 0, 0)
    assert rc == -1
    with pytest.raises(ValueError, match="null pointer"):
        _lib.check(rc, "pg_conv2d_taps")
    rc = lib.pg_causal_attn_fwd(1, 1, 1, 1, 1, 2, 1, 16, 128, 4, 0, 0, 0, 0, 0, 0)
    assert rc == -2  # head dim 128 unsupported: explicit error, no silent fallback
    with pytest.raises(ValueError, match="unsupported"):
        _lib.check(rc, "pg_causal_attn_fwd")


@pytest.mark.parametrize("name", _util.golden_names())
def test_state_dict_layout_matches_reference(name):
    import pytorch_generative_amd as pg

    g = _util.load_golden(name)
    model = getattr(pg.models, g["ctor"])(**g["kwargs"])
    want = {k: tuple(v.shape) for k, v in g["state0"].items() if k not in ("_c", "_h", "_w")}
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == want
    sd = dict(g["state0"])  # a checkpoint saved after training also holds the lazy buffers
    _, c, h, w = g["x"].shape
    sd.update({"_c": torch.tensor(c), "_h": torch.tensor(h), "_w": torch.tensor(w)})
    model.load_state_dict(sd, strict=True)
    assert (int(model._c), int(model._h), int(model._w)) == (c, h, w)
    for k, v in g["state0"].items():
        if k.endswith(".mask"):
            assert torch.equal(model.state_dict()[k], v), k  # causal masks bit-exact


@pytest.mark.parametrize("name", _util.vae_golden_names())
def test_vae_state_dict_layout_matches_reference(name):
    import pytorch_generative_amd as pg

    g = _util.load_golden(name)
    model = getattr(pg.models, g["ctor"])(**g["kwargs"])
    want = {k: tuple(v.shape) for k, v in g["state0"].items() if k not in ("_c", "_h", "_w")}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == want
    model.load_state_dict(g["state0"], strict=True)


def test_hip_path_has_no_cpu_fallback():
    import pytorch_generative_amd as pg

    conv = pg.nn.CausalConv2d(True, 1, 4, kernel_size=3, padding=1)
    with pytest.raises(RuntimeError, match="cuda"):
        conv(torch.zeros(1, 1, 8, 8))
    with pytest.raises(ValueError):
        pg.nn.Conv2d(1, 1, 3, stride=2)
    with pytest.raises(ValueError):
        pg.nn.GatedActivation(activation_fn=torch.relu)


def test_conv_spec_taps():
    from pytorch_generative_amd import ops

    s = ops.ConvSpec(2, 2, 1, 1)  # PixelSNAIL 2x2 pad 1 (cropped): taps (-1,-1)...(0,0)
    assert sorted(zip(s.f_dr, s.f_dc)) == [(-1, -1), (-1, 0), (0, -1), (0, 0)]
    s = ops.ConvSpec(3, 3, 1, 1, active=[(0, 0), (0, 1), (0, 2), (1, 0)])
    assert len(s.fwd_taps) == 4 and len(s.wg_taps) == 9
    assert list(zip(s.f_ndr, s.f_ndc))[0] == (1, 1)
    s = ops.ConvSpec(2, 1, 2, 0)  # gated vstack (k//2+1)x1, padding k//2+1 for k=3
    assert sorted(s.f_dr) == [-2, -1] and s.full_out(8, 8) == (11, 8)


def _gloo_worker(rank, world, port, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_generative_amd import parallel

    class _Opt:  # the two attributes FlatGradAllReduce needs, on CPU
        def __init__(self):
            self.flat_grad = torch.full((10,), float(rank + 1))
            self.flat_param = torch.full((10,), float(rank))
            self.prescale = None

        def set_grad_prescale(self, s):
            self.prescale = s

    opt = _Opt()
    red = parallel.FlatGradAllReduce(opt)
    red.broadcast_parameters(src=0)
    red.all_reduce()
    ok = (opt.prescale == 1.0 / world and torch.equal(opt.flat_param, torch.zeros(10))
          and torch.equal(opt.flat_grad, torch.full((10,), 3.0)))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert dict(ret) == {0: True, 1: True}


def test_flat_layout_places_declared_followers_back_to_back():
    """optim.plan_layout: 16-byte aligned slices, `_pg_follows` pairs adjacent (what lets
    nn.CausalAttention run its q and kv projections as one convolution over merged views)."""
    import torch
    from pytorch_generative_amd import nn as pg_nn, optim

    attn = pg_nn.CausalAttention(in_channels=16, n_heads=4, embed_channels=16, out_channels=16)
    extra = torch.nn.Parameter(torch.zeros(5))  # 5 elements: the next slice must be padded to 8
    params = [extra] + list(attn.parameters())
    offs, total = optim.plan_layout(params)
    by = {id(p): o for p, o in zip(params, offs)}
    assert by[id(extra)] == 0 and by[id(attn._q.weight)] == 8
    assert by[id(attn._kv.weight)] == by[id(attn._q.weight)] + attn._q.weight.numel()
    assert by[id(attn._kv.bias)] == by[id(attn._q.bias)] + attn._q.bias.numel()
    spans = sorted((o, o + p.numel()) for p, o in zip(params, offs))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "slices overlap"
    assert all(o % 4 == 0 for o in offs) and total >= spans[-1][1]
    # a CausalAttention with extra kv inputs (PixelSNAIL) declares nothing: plain aligned layout
    snail = pg_nn.CausalAttention(in_channels=8, embed_channels=4, out_channels=8, extra_input_channels=3)
    assert not hasattr(snail._kv.weight, "_pg_follows")


def test_adjacent_view_merges_only_contiguous_neighbours():
    import torch
    from pytorch_generative_amd import ops

    flat = torch.arange(40, dtype=torch.float32)
    a, b = flat[0:16].view(4, 4), flat[16:24].view(2, 4)
    merged = ops._adjacent_view(a, b, (6, 4))
    assert merged is not None and torch.equal(merged, flat[:24].view(6, 4))
    merged[5, 3] = -1.0  # zero copy: writes land in the flat buffer
    assert flat[23] == -1.0
    assert ops._adjacent_view(a, flat[20:28].view(2, 4), (6, 4)) is None   # gap
    assert ops._adjacent_view(a, torch.zeros(2, 4), (6, 4)) is None        # different storage


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r02_bench_final.json is a real `python bench.py` line: the keys the driver and the
    judge read (bench contract + roofline + cpu_baseline) must all be there and self-consistent."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                        "r02_bench_final.json")
    with open(path) as f:
        d = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    # value = images of all GPUs / time of exactly `steps` steps
    per_step = d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)
    assert abs(per_step - d["value"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0
    ps = d["pixel_snail"]
    assert ps["images_per_s"] > 0 and "roofline" in ps and "reference_default_batch_128" in ps


def test_no_kernel_spills_or_scratch():
    """Every kernel of the built library: no spilled vector register, no scratch memory (tools/kernel_resources.py reads
    the code objects' metadata; build.py enforces the same at link time). Skipped when the objects are not there."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, "pytorch-generative_amd", "build")):
        pytest.skip("no object files (run __graft_entry__.build())")
    spec = importlib.util.spec_from_file_location("kr", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.table()
    assert len(rows) > 300
    bad = [r for r in rows if r["vgpr_spill"] or r["scratch_B"]]
    assert not bad, kr.render(bad)


def test_guard_allocator_library_exports():
    """tests/guard/libpg_guard.so (the canary allocator of `PG_GUARD=1 pytest -m gpu`) builds and exports the entry
    points torch's pluggable-allocator hook and the conftest fixture bind."""
    import ctypes

    import guard

    so = guard.build()
    lib = ctypes.CDLL(so)
    for name in ("pg_guard_malloc", "pg_guard_free", "pg_guard_check_all", "pg_guard_violations", "pg_guard_report",
                 "pg_guard_live", "pg_guard_capture"):
        assert hasattr(lib, name), name


def test_attention_dims_native_matches_the_kernels_table():
    """ops.attention_dims_native mirrors the instantiation rules of csrc/attention*.hip (include/pg_hip.h)."""
    from pytorch_generative_amd import ops

    assert ops.attention_dims_native(4, 4, 49) and ops.attention_dims_native(2, 2, 49)          # VALU / m44, any L
    assert ops.attention_dims_native(4, 32, 1024) and ops.attention_dims_native(16, 16, 50)
    assert ops.attention_dims_native(32, 32, 784) and ops.attention_dims_native(64, 64, 64)     # matrix-core k4
    assert not ops.attention_dims_native(64, 64, 49)      # L % 16 != 0 -> padded by ops.causal_attention
    assert not ops.attention_dims_native(24, 40, 48) and not ops.attention_dims_native(8, 20, 256)
    assert not ops.attention_dims_native(16, 32, 49)


def _arange_like_kernel(n):
    """`arange_like_torch_cpu` of elementwise.hip restated with exact rational arithmetic: every fma is one correctly
    rounded double operation (Fraction -> float rounds to nearest even), the float32 store a second rounding."""
    from fractions import Fraction

    import numpy as np

    step = 1.0 / float(n)  # the double the kernel computes
    fma = lambda a, b, c: float(Fraction(a) * Fraction(b) + Fraction(c))  # noqa: E731
    nvec = (n // 16) * 16
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        if i < nvec:
            b0 = (i // 8) * 8
            base = np.float32(fma(float(b0), step, -0.5))
            out[i] = np.float32(fma(float(i - b0), step, float(base)))
        else:
            out[i] = np.float32(fma(float(i), step, -0.5))
    return out


def test_positional_encoding_formula_reproduces_the_reference_fixture():
    """The rounding recipe `posenc_kernel` implements (ATen's vectorised CPU arange: 8-wide vectors from a float32 base
    for the first floor(n / 16) * 16 elements, scalar fma for the rest), evaluated WITHOUT a GPU, equals the planes the real
    reference produced bit for bit (tests/golden/posenc.pt) — so the GPU test's strict `torch.equal` does not depend on
    which host the GPU box has; plus a sweep of sizes against the oracle on this host when it is an AVX-512 one."""
    import torch

    gold = _util.load_golden("posenc")
    for shape, enc in gold["cases"].items():
        h, w = shape[2], shape[3]
        rows = torch.from_numpy(_arange_like_kernel(h))
        cols = torch.from_numpy(_arange_like_kernel(w))
        assert torch.equal(enc[0, 0], rows[:, None].expand(h, w)), shape
        assert torch.equal(enc[0, 1], cols[None, :].expand(h, w)), shape
    if torch.backends.cpu.get_cpu_capability() == gold["cpu_capability"]:
        for n in list(range(2, 70)) + [96, 100, 127, 128, 200, 255, 256, 511, 600]:
            assert torch.equal(torch.from_numpy(_arange_like_kernel(n)), torch.arange(-0.5, 0.5, 1 / n)[:n]), n


def test_bench_self_spawn_watches_all_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` as a bare command (bench._self_spawn): every rank gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* on 127.0.0.1, rank 0's stdout is forwarded, and a rank that dies does not leave the command hanging on the
    survivors (they would sit in a collective forever): non-zero exit after a grace period. Ranks here are a stub script —
    no GPU involved."""
    import importlib.util
    import sys
    import time

    spec = importlib.util.spec_from_file_location("pg_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "open(os.path.join(os.path.dirname(__file__), f'seen_{r}_of_{w}'), 'w').write(' '.join(sys.argv[1:]))\n"
        "mode = sys.argv[1]\n"
        "if mode == 'ok': sys.exit(0)\n"
        "if r == 1: sys.exit(3)\n"      # mode 'die': rank 1 fails at once, the others would wait forever
        "time.sleep(120)\n")
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setenv("PG_BENCH_RANK_GRACE_S", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "ok", "--gpus", "3"])
    assert bench._self_spawn(3) == 0
    assert sorted(p.name for p in tmp_path.glob("seen_*")) == ["seen_0_of_3", "seen_1_of_3", "seen_2_of_3"]
    assert (tmp_path / "seen_2_of_3").read_text() == "ok --gpus 3"
    monkeypatch.setattr(sys, "argv", ["bench.py", "die", "--gpus", "3"])
    t0 = time.monotonic()
    assert bench._self_spawn(3) != 0
    assert time.monotonic() - t0 < 30, "a dead rank must not leave the launcher waiting for the survivors"


# Which kernel family a convolution shape is sent to (csrc/conv_mfma.hip pg_conv_mfma_supported: a host function, no GPU needed).
# 0 = the VALU tap kernel, 1 = fp32 MFMA fragments (conv_mfma_kernel), 2 = bf16x3 fragments (conv_b3*). The shapes are the layers the
# eight bench workloads launch (DESIGN.md section 3's kernel table); a change here moves a model onto another kernel, so it should be a
# decision (profiles/README.md records each one: e.g. round 5 item 7 for the 3-channel input layers), not a side effect.
CONV_ROUTING = [
    ("PixelSNAIL input 3 -> 64, 4 causal taps, 32x32", (3, 64, 4, 32, 32, 32, 1, 2), 1),
    ("PixelSNAIL merged q/k/v projection 69 -> 40, 1x1", (69, 40, 1, 32, 32, 32, 0, 0), 1),
    ("PixelSNAIL 2x2 64 -> 64", (64, 64, 4, 32, 32, 32, 1, 1), 2),
    ("PixelSNAIL 1x1 64 -> 128", (64, 128, 1, 32, 32, 32, 0, 0), 2),
    ("PixelCNN 7x7 input 1 -> 32 (24 taps: more than the fragment kernels' tap table)", (1, 32, 24, 28, 28, 28, 3, 6), 0),
    ("ImageGPT 3x3 input 1 -> 16", (1, 16, 4, 28, 28, 28, 1, 2), 0),
    ("GatedPixelCNN vertical 2x3 128 -> 256", (128, 256, 6, 32, 32, 32, 1, 2), 2),
    ("beta-VAE 4x4 / stride 2 input layer as a 2x2 phase convolution 3 -> 64", (3, 64, 4, 32, 32, 32, 1, 1), 1),
    ("VD-VAE 3x3 32 -> 32 on 64x64", (32, 32, 9, 64, 64, 64, 2, 2), 2),
    ("PixelCNN++ 2x3 160 -> 160", (160, 160, 6, 32, 32, 32, 1, 2), 2),
    ("Cin 24 (a multiple of 8, not of 16)", (24, 128, 1, 32, 32, 32, 0, 0), 2),
    ("Cout 56 (a partial last 16-channel tile)", (64, 56, 1, 28, 28, 28, 0, 0), 2),
    ("rows wider than a 256-pixel tile", (64, 64, 1, 4, 320, 320, 0, 0), 0),
    ("a 2x2 image (VD-VAE's lowest levels)", (64, 64, 9, 2, 2, 2, 2, 2), 0),
    ("2 input channels", (2, 64, 4, 32, 32, 32, 1, 1), 0),
    ("4 input channels but only 16 outputs", (4, 16, 4, 32, 32, 32, 1, 1), 0),
]


@pytest.mark.parametrize("what,shape,fmt", CONV_ROUTING, ids=[c[0] for c in CONV_ROUTING])
def test_convolution_routing_table(lib, what, shape, fmt):
    assert lib.pg_conv_mfma_supported(*shape) == fmt, what
    if fmt:  # the fragment buffer the caller must allocate for that format is non-empty and 16-byte granular
        cin, cout, taps = shape[0], shape[1], shape[2]
        n = lib.pg_conv_frag_floats(cin, cout, taps, fmt)
        assert n > 0 and n % 4 == 0


def test_lds_zero_fills_are_ordered_before_the_first_commit():
    """Source-level guard for the race of round 5 (profiles/README.md item 16): every kernel that zero-fills an LDS tile once and relies on the
    halo staying zero must pass a workgroup barrier before another thread may commit into that tile. The GPU tier has the behavioural check
    (tests/test_gpu_shared_device.py); this one fails on the CPU as soon as somebody removes a barrier."""
    csrc = os.path.join(ROOT, "pytorch-generative_amd", "csrc")

    def between(path, start, end):
        text = open(os.path.join(csrc, path)).read()
        i = text.index(start)
        return text[i:text.index(end, i)]

    # fp32-MFMA kernel: zero fill ... issue(0) ... barrier ... commit(0)
    assert "__syncthreads();" in between("conv_mfma.hip", "lds[a.buf_stride + i] = 0.f;", "PG_MF_COMMIT_ALL(0, 0)")
    # bf16x3 kernels: zero fill ... barrier ... (first issue / commit)
    assert "__syncthreads();" in between("conv_b3_kernels.h", "i < a.cgs * 3 * a.plane16; i += THREADS) lds16[i]", "PG_B3_COMMIT_ALL()")
    assert "__syncthreads();" in between("conv_b3_kernels.h", "i < 2 * xbuf16; i += THREADS) lds16[i]", "PG_P_COMMIT_X(0, 0)")
    # VALU tap kernel and fp32 weight-gradient kernel: the chunk / tile loop opens with a barrier
    assert "__syncthreads();" in between("conv_direct.hip", "i < a.CIB * a.ch_stride; i += blockDim.x) lds[i] = 0.f;", "pg_stage_rows_vec4<ACT>")
    assert "__syncthreads();" in between("conv_wgrad.hip", "i < lds_floats; i += WG_THREADS) lds[i] = 0.f;", "PG_WG_COMMIT_X(PG_ACT_RELU) break;")
    # the overlapped 16-wave kernel (round 6)
    assert "__syncthreads();" in between("conv_b3q_kernel.h", "i < NXT * 2 * xbuf16; i += B3Q_THREADS) lds16[i]", "PG_Q_COMMIT_X(0)")


def test_lds_dma_slabs_are_waited_for_before_the_barrier_that_publishes_them():
    """Source-level guard for the weight slabs moved by LDS-DMA (global_load_lds_dwordx4 issued by inline asm: INVISIBLE to the compiler's
    waitcnt model, advisor finding of round 5): between every DMA issue inside a step loop and the workgroup barrier that lets other waves
    read the slab there must be an explicit s_waitcnt vmcnt. The behavioural checks are the convolution parity cases (destinations up to
    150 KB into LDS in conv_b3q_kernel: every launch of tests/test_gpu_ops.py's 6-tap cases) and the twin-process tier."""
    csrc = os.path.join(ROOT, "pytorch-generative_amd", "csrc")
    for path, dma, stop in (("conv_b3_kernels.h", "if (more) PG_B3_WGLDS(step + 1)", "if (more) __syncthreads();"),
                            ("conv_b3_kernels.h", "PG_B3_WGLDS(0)", "__syncthreads();"),
                            ("conv_b3q_kernel.h", "PG_Q_DMA(nj_, same_ ? ks + 1 : 0, stage ^ 1)", "if (more_q) __syncthreads();"),
                            ("conv_b3q_kernel.h", "PG_Q_DMA(0, 0, 0)", "__syncthreads();")):
        text = open(os.path.join(csrc, path)).read()
        i = text.index(dma)
        seg = text[i:text.index(stop, i)]
        assert re.search(r's_waitcnt vmcnt\(', seg), f"{path}: no explicit vmcnt wait between `{dma}` and the barrier"


# ---- planners without a GPU ------------------------------------------------------------------------------------------
# Without a device every launch fails with hipErrorNoDevice (100) AFTER the host-side planning (tile geometry, channel
# chunks, LDS budget, staging slots) has accepted the problem; a problem the planner refuses comes back as a negative
# PG_E* code before any launch. So on the CPU box the planners can be swept over thousands of shapes: whatever the routing
# query promises, the launch must accept. (Pointers are fake and never dereferenced: these tests must not run where a
# launch would succeed.)
_NO_DEVICE = 100
_no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="sweeps the planners with fake pointers: CPU box only")


def _fake_ptr():
    import ctypes

    import numpy as np
    buf = np.zeros(1 << 12, dtype=np.float32)
    return buf, ctypes.c_void_p((buf.ctypes.data + 255) // 256 * 256)


def _reached_launch(lib, rc):
    return rc == _NO_DEVICE or b"launch failed" in lib.pg_last_error()


@_no_gpu
@pytest.mark.parametrize("seed", range(4))
def test_routing_never_promises_a_shape_the_launch_refuses(lib, seed):
    """Round 5: pg_conv_mfma_supported sent 5x5 / 7x7 kernels with 13+ active taps to the fp32-MFMA kernel, whose planner then
    refused them ("exceeds the staging slots") - a ValueError where the VALU tap kernel would have worked. The query now runs the
    launch's own geometry function; this sweep holds the two together for forward and data gradient, and checks that the tap
    kernel takes everything the query does not route to the matrix cores."""
    import random

    from pytorch_generative_amd import _lib
    ia = _lib.int_array
    keep, p = _fake_ptr()
    r = random.Random(seed)
    kernels = [(1, 1), (2, 2), (3, 3), (2, 3), (1, 3), (3, 1), (2, 1), (1, 2), (5, 5), (4, 4), (7, 7), (3, 5)]
    checked = 0
    for _ in range(1500):
        kh, kw = r.choice(kernels)
        ph, pw = r.choice([(0, 0), (kh // 2, kw // 2), (kh - 1, kw - 1), (kh - 1, kw // 2)])
        taps = [(u, v) for u in range(kh) for v in range(kw)]
        mode = r.choice(["all", "A", "B", "random"])
        if mode == "A":
            taps = [(u, v) for u, v in taps if u < kh // 2 or (u == kh // 2 and v < kw // 2)] or taps
        elif mode == "B":
            taps = [(u, v) for u, v in taps if u < kh // 2 or (u == kh // 2 and v <= kw // 2)]
        elif mode == "random":
            taps = [t for t in taps if r.random() < 0.6] or taps
        cin = r.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32, 33, 40, 48, 64, 66, 69, 96, 100, 128, 160, 192, 256, 320, 512])
        cout = r.choice([1, 2, 3, 8, 10, 16, 24, 32, 36, 40, 56, 64, 72, 96, 100, 128, 160, 192, 256, 320, 512])
        ih = r.choice([1, 2, 3, 4, 5, 7, 8, 12, 14, 16, 20, 28, 32, 33, 48, 64, 100, 128])
        iw = r.choice([1, 2, 3, 4, 6, 8, 12, 14, 16, 20, 28, 32, 36, 48, 64, 100, 128, 256, 300])
        n = r.choice([1, 2, 3, 7, 16, 33])
        oh, ow = ih + 2 * ph - kh + 1, iw + 2 * pw - kw + 1
        if oh < 1 or ow < 1:
            continue
        t = len(taps)
        hr = max(u for u, _ in taps) - min(u for u, _ in taps)
        hc = max(v for _, v in taps) - min(v for _, v in taps)
        problems = (  # (K channels, IH, IW, M channels, OH, OW, tap rows, tap columns)
            (cin, ih, iw, cout, oh, ow, [u - ph for u, _ in taps], [v - pw for _, v in taps]),  # forward
            (cout, oh, ow, cin, ih, iw, [ph - u for u, _ in taps], [pw - v for _, v in taps]),  # data gradient
        )
        for kc, h_in, w_in, m, h_out, w_out, dr, dc in problems:
            fmt = lib.pg_conv_mfma_supported(kc, m, t, h_out, w_out, w_in, hr, hc)
            if fmt:
                rc = lib.pg_conv2d_mfma_ex(p, p, None, None, p, n, kc, h_in, w_in, m, h_out, w_out, t, ia(dr), ia(dc), 0, None, 0, 0,
                                           fmt, None, 0, 0, None)
            else:
                rc = lib.pg_conv2d_taps(p, p, None, None, p, n, kc, h_in, w_in, m, h_out, w_out, t, ia(dr), ia(dc), 0, None, 0, None)
            assert _reached_launch(lib, rc), (fmt, rc, lib.pg_last_error(), (kh, kw, ph, pw, t, cin, cout, ih, iw, n))
            checked += 1
    assert checked > 2000
    del keep


@_no_gpu
def test_weight_gradient_planner_takes_every_layer_of_the_bench_workloads(lib):
    """pg_conv2d_wgrad picks among five kernels; each has LDS / slot limits (the widest row it takes shrinks with the kernel size:
    DESIGN.md section 6, "limits"). Every convolution of the eight bench workloads, at every resolution its model runs it, must
    pass the planner - at batch 1 and at batch 64."""
    import bench
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import _lib
    ia = _lib.int_array
    keep, p = _fake_ptr()
    checked = 0
    for name, w in bench.WORKLOADS.items():
        model = getattr(pg.models, w["ctor"])(**w["kw"])
        side = w["chw"][1]
        sides = [side] if name in ("image_gpt", "image_gpt_repro", "pixel_snail", "gated_pixel_cnn", "pixel_cnn") else \
            [s for s in (64, 32, 16, 8, 4, 2, 1) if s <= side]
        seen = set()
        for label, mod in model.named_modules():
            if not hasattr(mod, "_conv_spec") or getattr(mod, "weight", None) is None or mod.weight.dim() != 4:
                continue
            sp = mod._conv_spec()
            cout, cin = mod.weight.shape[:2]
            taps = sp.wg_taps
            for s in sides:
                oh, ow = sp.full_out(s, s)
                oh, ow = min(oh, s), min(ow, s)  # the causal layers crop their output back to the input size
                key = (cin, cout, sp.kh, sp.kw, sp.pad_h, sp.pad_w, len(taps), s, oh, ow)
                if key in seen or oh < 1 or ow < 1:
                    continue
                seen.add(key)
                ws = lib.pg_conv2d_wgrad_workspace_floats(cout, cin, len(taps))
                for n in (1, 64):
                    rc = lib.pg_conv2d_wgrad(p, p, p, p, n, cin, s, s, cout, oh, ow, sp.kh, sp.kw, len(taps), sp.w_dr, sp.w_dc,
                                             sp.w_u, sp.w_v, 0, p, ws, None)
                    assert _reached_launch(lib, rc), (name, label, key, n, rc, lib.pg_last_error())
                    checked += 1
    assert checked > 100
    del keep


@_no_gpu
def test_attention_entry_points_accept_exactly_the_native_dims(lib):
    """ops.attention_dims_native (what the Python host sends to the kernels unpadded) against the planners of pg_causal_attn_fwd / _bwd:
    accepted if and only if native, for every head-dim / length / head-count / batch combination of the grid."""
    from pytorch_generative_amd import ops
    keep, p = _fake_ptr()
    checked = 0
    for dk in (1, 2, 3, 4, 5, 8, 12, 16, 32, 64, 128):
        for dv in (1, 2, 4, 8, 16, 32, 64, 128):
            for seq in (1, 2, 5, 16, 17, 49, 64, 100, 784, 1024, 1040):
                for heads, n in ((1, 1), (2, 3), (4, 1024)):
                    for strict in (0, 1):
                        native = ops.attention_dims_native(dk, dv, seq)
                        qb, vb = heads * dk * seq, heads * dv * seq
                        rc = lib.pg_causal_attn_fwd(p, p, p, p, p, n, heads, seq, dk, dv, qb, qb, vb, vb, strict, None)
                        assert _reached_launch(lib, rc) == native, ("fwd", dk, dv, seq, heads, n, rc, lib.pg_last_error())
                        rc = lib.pg_causal_attn_bwd(p, p, p, p, p, p, p, p, p, p, n, heads, seq, dk, dv, qb, qb, vb, vb, vb, qb, qb, vb,
                                                    strict, None)
                        assert _reached_launch(lib, rc) == native, ("bwd", dk, dv, seq, heads, n, rc, lib.pg_last_error())
                        checked += 1
    assert checked > 5000
    del keep


@_no_gpu
def test_shape_agnostic_entry_points_take_any_size(lib):
    """LayerNorm, gates, pooling / upsampling / phase split, Gaussian heads, the losses: no shape of the grid is refused."""
    import ctypes
    import random
    keep, p = _fake_ptr()
    r = random.Random(0)
    for _ in range(600):
        n = r.choice([1, 2, 3, 17, 64, 1024])
        c = r.choice([1, 2, 3, 4, 8, 16, 17, 32, 48, 64, 100, 128, 256, 512])
        seq = r.choice([1, 2, 3, 4, 16, 49, 64, 196, 784, 1024, 4096])
        h, w = r.choice([1, 2, 3, 4, 7, 8, 16, 32]), r.choice([1, 2, 3, 4, 7, 8, 16, 32])
        ws = lib.pg_nchw_layernorm_bwd_workspace_floats(n, c, seq)
        calls = [
            ("layernorm_fwd", lib.pg_nchw_layernorm_fwd(p, p, p, p, p, p, n, c, seq, ctypes.c_float(1e-5), None)),
            ("layernorm_bwd", lib.pg_nchw_layernorm_bwd(p, p, p, p, p, p, p, p, n, c, seq, p, ws, None)),
            ("layernorm_bwd_res", lib.pg_nchw_layernorm_bwd_res(p, p, p, p, p, p, p, p, p, n, c, seq, p, ws, None)),
            ("gated_fwd", lib.pg_gated_fwd(p, p, n, c, seq, 0, None)),
            ("gated_bwd", lib.pg_gated_bwd(p, p, p, n, c, seq, 1, None)),
            ("avgpool2_fwd", lib.pg_avgpool2_fwd(p, p, n * c, h, w, None)),
            ("avgpool2_bwd", lib.pg_avgpool2_bwd(p, p, n * c, h, w, None)),
            ("upsample2_fwd", lib.pg_upsample2_fwd(p, p, n * c, h, w, None)),
            ("upsample2_bwd", lib.pg_upsample2_bwd(p, p, n * c, h, w, None)),
            ("phase_split2", lib.pg_phase_split2(p, p, n * c, h, w, 0, None)),
            ("gauss_head_fwd", lib.pg_gauss_head_fwd(p, p, p, p, p, n, c, seq, 2 * c * seq, 2 * c * seq, 1, None)),
            ("gauss_head_bwd", lib.pg_gauss_head_bwd(p, p, p, p, p, p, p, n, c, seq, 2 * c * seq, 2 * c * seq, 1, None)),
            ("dmol_fwd", lib.pg_dmol_fwd(p, p, p, n, r.choice([1, 5, 10]), seq, None)),
            ("bce_logits_fwd", lib.pg_bce_logits_fwd(p, p, p, n, c * seq, None)),
        ]
        for name, rc in calls:
            assert rc == _NO_DEVICE, (name, n, c, seq, h, w, rc)
    del keep


def test_bench_watchdog_aborts_a_rank_that_blocks():
    """bench.py's capture watchdog (a rank that BLOCKS in warm-up / capture next to a live communicator must end the job, not hang
    it): armed with a 1-second limit around a block that sleeps, the process must exit with code 3 and say why."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench._Watchdog('a capture that blocks', 0):\n    time.sleep(30)\nprint('survived')") % root
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PG_BENCH_CAPTURE_TIMEOUT_S="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr.decode()[-500:])
    assert b"did not finish within 1 s" in p.stderr and b"survived" not in p.stdout


def test_ops_package_reexports_every_operator_name():
    """Round 6 split ops.py into the package ops/ (one module per kernel family): every public operator and every helper the
    tests / tools / nn modules reach through `ops.<name>` must still resolve on the package."""
    from pytorch_generative_amd import ops

    for name in ("conv2d_taps", "conv2d_pair", "ConvSpec", "causal_attention", "causal_attention_qkv", "set_deterministic",
                 "gated_activation", "nchw_layernorm", "nchw_layernorm_skip", "image_positional_encoding", "mul_inplace_",
                 "bce_with_logits_sum_mean", "dmol_loss_sum_mean", "elbo_terms", "gaussian_head_unit", "gaussian_head_pair",
                 "avg_pool2", "upsample2_nearest", "phase_split", "phase_split4", "phase_merge", "phase_merge4", "phase_weights",
                 "split_in_channels", "concat_elu", "concat_channels", "subsample2", "zero_insert2", "fanout", "sum_vectors",
                 "zeros", "zeros_like", "gpt_block_head", "gpt_block_tail", "gpt_block_supported", "mlp_gelu", "RowDecode",
                 "add", "add_broadcast_batch", "relu", "elu", "gelu", "merge_qkv_weight", "new_block_chain",
                 "_use_mfma", "_pack_frag", "_adjacent_view", "_Act", "_ACT_IDS", "_sink", "_chk", "_dense_per_image",
                 "ACT_NONE", "ACT_RELU", "ACT_ELU", "ACT_GELU", "ACT_ELU_OUT", "GATE_TANH", "GATE_IDENTITY",
                 "CONV_FMT_B3", "FUSE_SKIP", "FUSE_PAIR", "FUSE_BLOCK", "DEFER_BLOCK_REDUCE"):
        assert hasattr(ops, name), name
    assert ops.conv2d_taps.__module__ == "pytorch_generative_amd.ops.conv"
    assert ops.causal_attention.__module__ == "pytorch_generative_amd.ops.attention"


def test_bench_reads_the_committed_counter_passes():
    """bench.py's roofline fields that come from the tracked rocprofv3 --pmc passes (profiles/r06_*.json): bytes, source file,
    matrix-pipe busy fraction, VALU instructions per MFMA, and the achieved HBM rate derived from a launch time."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    traffic, src, ctr = bench.measured_traffic(1024, "attn_bwd_m44_kernel", with_counters=True)
    assert src == "profiles/r06_traffic.json" and 4.0e8 < traffic < 4.5e8
    f = bench._counter_fields(traffic, ctr, 0.62)
    assert 0.45 < f["mfma_busy_frac"] < 0.56 and 600 < f["hbm_gbps_achieved"] < 750 and f["valu_per_mfma"] > 1
    t2, src2 = bench.measured_traffic(1024, "attn_bwd_m44_kernel")
    assert (t2, src2) == (traffic, src)
    assert bench.measured_traffic(7, "attn_bwd_m44_kernel", with_counters=True) == (None, None, None)
    dom = bench._dominant_kernel("gated_pixel_cnn")
    assert dom["source"] == "profiles/r06_gated_pixel_cnn_kernel_stats.csv" and dom["share_of_kernel_time"] > 0.4


def test_deferred_block_reductions_must_be_flushed_before_a_step():
    from pytorch_generative_amd import ops

    chain = ops.new_block_chain()
    ops.assert_no_pending_block_reductions()
    chain["jobs"].append(("head_ws", "tail_ws", [0] * 14, None))
    with pytest.raises(RuntimeError, match="never flushed"):
        ops.assert_no_pending_block_reductions()
    chain["jobs"].clear()
    ops.assert_no_pending_block_reductions()
