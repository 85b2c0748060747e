"""Canary allocator harness for the GPU tests (TEST INFRASTRUCTURE — the package never imports this).

`PG_GUARD=1 python -m pytest tests -m gpu` routes every torch device allocation through
tests/guard/libpg_guard.so (pg_guard_alloc.hip): each tensor is its own device allocation between two 64 KB margins of
canary bytes, its contents poisoned with NaN bytes. An out-of-bounds WRITE of any kernel is reported by the per-test
canary check that tests/conftest.py installs (offsets relative to the tensor, the bytes written); a read of
never-written memory shows up as NaN in the test's own comparison.

    PG_GUARD_ALIGN=512|16       rounding of sizes: 512 = what torch's caching allocator guarantees, 16 = strict
    PG_GUARD_MARGIN_KB=64       canary bytes on each side
    PG_GUARD_POISON=0           do not fill fresh tensors with NaN bytes
    PG_GUARD_POISON_BYTE=0x5C   poison with 2.5e17 (huge but FINITE) instead of NaN: max / select operations (ReLU, masks) swallow
                                a NaN, a huge finite value survives them and shows up in the comparison
    PG_GUARD_ARENA_MB=32768     arena that serves allocations made DURING a hipGraph capture (never recycled: a captured graph
                                keeps its addresses); memory of the arena is virgin, which is how a read of stale graph memory
                                shows (profiles/README.md, round 4, item 12)
    PG_GUARD_MODE=vmm           experimental guard PAGES (out-of-bounds reads fault); unreliable on ROCm 7.2, see the .hip
"""

import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libpg_guard.so")
SRC = os.path.join(HERE, "pg_guard_alloc.hip")
_lib = None


def build(force=False):
    if not force and os.path.exists(SO) and os.path.getmtime(SO) > os.path.getmtime(SRC):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # several processes may get here at once (pytest-xdist workers, two PG_GUARD runs): each links into its own temporary
    # file and renames it into place atomically, so that nobody ever dlopens a half-written library
    tmp = f"{SO}.{os.getpid()}.tmp"
    try:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", SRC, "-o", tmp], check=True)
        os.replace(tmp, SO)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return SO


def enabled():
    return os.environ.get("PG_GUARD") == "1"


def install():
    """Makes the guard allocator torch's device allocator. Must run before the first device allocation."""
    global _lib
    import torch

    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        build()  # lazily: __graft_entry__.build() normally ships it, but the product build does not depend on it
    alloc = torch.cuda.memory.CUDAPluggableAllocator(SO, "pg_guard_malloc", "pg_guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    _lib = ctypes.CDLL(SO)
    _lib.pg_guard_report.restype = ctypes.c_char_p
    _lib.pg_guard_live.restype = ctypes.c_long
    # tell the allocator when a hipGraph capture is in progress (it must then neither allocate, synchronise nor free):
    # every capture of this code base goes through torch.cuda.graph
    lib = _lib
    enter, leave = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__

    def _enter(self):
        torch.cuda.synchronize()
        lib.pg_guard_capture(1)
        try:
            return enter(self)
        except BaseException:
            lib.pg_guard_capture(0)
            raise

    def _exit(self, *exc):
        try:
            return leave(self, *exc)
        finally:
            lib.pg_guard_capture(0)

    torch.cuda.graph.__enter__, torch.cuda.graph.__exit__ = _enter, _exit
    return _lib


def check_all():
    """Verifies the canaries of every live tensor; returns (violations so far, report text)."""
    if _lib is None:
        return 0, ""
    n = _lib.pg_guard_check_all()
    return n, _lib.pg_guard_report().decode()
