"""Builds libpg_hip.so (the gfx950 kernels + C-ABI) in-tree with hipcc.

Usage: python build.py [--force]
The .so lands in pytorch_generative_amd/lib/ (git-ignored, but it travels with gpurun).
"""

import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PG_ABLATE=1: the timing-only ablation build (-DPG_ABLATE: PG_B3_DBG / PG_WB_DBG are compiled in) as a SEPARATE
# library, lib/libpg_hip_ablate.so, for tools/exp — the production library never contains those switches
ABLATE = os.environ.get("PG_ABLATE") == "1"
# PG_VARIANT=<name>: an EXPERIMENT build with -DPG_<NAME> (e.g. PG_VARIANT=ab -> -DPG_AB: the kernels' A/B switches live, common.h) as a
# separate library lib/libpg_hip_<name>.so; tests / bench pick it up with PG_HIP_LIB=<path> (pytorch_generative_amd/_lib.py)
VARIANT = os.environ.get("PG_VARIANT", "")
_TAG = "ablate" if ABLATE else VARIANT
OBJ = os.path.join(HERE, "build_" + _TAG if _TAG else "build")
LIB_DIR = os.path.join(HERE, "pytorch_generative_amd", "lib")
LIB = os.path.join(LIB_DIR, f"libpg_hip_{_TAG}.so" if _TAG else "libpg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fvisibility=hidden",
    "-munsafe-fp-atomics",  # global_atomic_add_f32 instead of a CAS loop
    # MFMA results in architectural VGPRs, never in AGPRs: every kernel here consumes its accumulators on the VALU (exp, scaling,
    # epilogues), and with the default (AGPR destinations picked by the allocator) attention_k4 / conv_wgrad / gpt_block carried up
    # to 64 v_accvgpr_read/write per loop iteration plus the registers to stage them: 103 kernels use fewer registers with this
    # (29 gain a wave per SIMD, e.g. attn_fwd_k4_kernel<1, 2, 4> 148 -> 128), none spills (profiles/README.md, round 4, item 13)
    "-mllvm", "-amdgpu-mfma-vgpr-form=1",
    "-Wall",
    "-Wno-unused-function",
] + (["-DPG_ABLATE"] if ABLATE else []) + ([f"-DPG_{VARIANT.upper()}"] if VARIANT else [])
# PG_EXTRA_FLAGS="..." (only together with PG_VARIANT): additional compiler flags of an experiment build, e.g. a scheduler strategy
if VARIANT and os.environ.get("PG_EXTRA_FLAGS"):
    FLAGS += os.environ["PG_EXTRA_FLAGS"].split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "pg_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if (
        not force
        and os.path.exists(obj)
        and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _deps_mtime())
    ):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", srcp, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    linked = rebuilt or force or not os.path.exists(LIB)  # evaluated BEFORE linking: a library removed by a failed check
    # is re-linked from up-to-date objects on the next call and must be checked again
    if linked:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        if verbose:
            print(f"[build] linked {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"[build] {LIB} up to date")
    if linked and not ABLATE:
        _check_resources(verbose)
    return LIB


def _check_resources(verbose):
    """No kernel of the production library may spill a vector register or use scratch memory: the register /
    scratch / LDS table of every kernel is read back from the code objects (tools/kernel_resources.py) and the build
    FAILS on the first spilled VGPR (PG_ALLOW_SPILLS=1 turns the failure into a report, for experiments)."""
    import importlib.util

    path = os.path.join(HERE, "..", "tools", "kernel_resources.py")
    if not os.path.exists(path):
        return
    spec = importlib.util.spec_from_file_location("pg_kernel_resources", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.OBJ = OBJ  # the objects of THIS build (a PG_VARIANT build is judged on build_<variant>/, not on build/)
    try:
        rows = mod.check(max_spill=0, verbose=verbose)
        scratch = [r for r in rows if r["scratch_B"]]
        if scratch:
            raise RuntimeError("kernels using scratch memory:\n" + mod.render(scratch))
    except RuntimeError as e:
        if os.environ.get("PG_ALLOW_SPILLS") == "1":
            sys.stderr.write(f"[build] WARNING (PG_ALLOW_SPILLS=1): {e}\n")
        else:
            os.remove(LIB)  # a library with spilling kernels must not be picked up by accident
            raise


if __name__ == "__main__":
    build(force="--force" in sys.argv)
