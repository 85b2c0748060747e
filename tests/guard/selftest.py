"""Self-test of the guard allocator on the GPU (python tests/guard/selftest.py): clean tensors raise no violation,
a deliberate 64-byte write in front of / behind a tensor is reported with the right offsets, and (child process)
a read past the guarded edge faults instead of returning."""

import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "pytorch-generative_amd"), ROOT]

import guard  # noqa: E402


def main():
    import torch

    side = os.environ.get("PG_GUARD_SIDE", "end")
    lib = guard.install()
    from pytorch_generative_amd import _lib

    pg = _lib.load()
    st = ctypes.c_void_p(0)
    # the copy engines on guarded pointers (the tensor sits at an offset inside its mapping)
    h = torch.arange(1000, dtype=torch.float32)
    dv = h.to("cuda")
    assert float(dv.sum()) == float(h.sum()), "H2D copy into a guarded tensor is wrong"
    assert torch.equal((dv * 1).cpu(), h) and torch.equal(dv.cpu(), h), "D2H copy out of a guarded tensor is wrong"
    small = torch.tensor([1.0, 2.0, 3.0]).to("cuda")
    assert small.cpu().tolist() == [1.0, 2.0, 3.0] and float(small.sum()) == 6.0
    z = torch.ones(6, device="cuda").zero_()
    assert z.cpu().tolist() == [0.0] * 6
    big = torch.arange(3_000_000, dtype=torch.float32)
    assert torch.equal(big.to("cuda").cpu(), big) and float((big.to("cuda") * 0 + 1).sum()) == 3_000_000.0
    print("copies in / out of guarded tensors: ok")
    t = torch.empty(6, device="cuda")
    assert not bool(torch.isfinite(t).any()), "fresh tensors are not poisoned"
    n, rep = guard.check_all()
    print("after empty(6):", n, rep.strip())
    t.zero_()
    a = torch.ones(6, device="cuda")
    b = torch.randn(1000, device="cuda") * 2
    c = torch.zeros((), device="cuda", dtype=torch.int64)
    d = (a + 1).sum()
    torch.cuda.synchronize()
    n1, rep = guard.check_all()
    print("after a few ATen ops:", n1, rep.strip()[-400:])
    assert n1 == 0, "violations without any out-of-bounds access: the harness itself is wrong"
    # deliberate damage on the canary side: pg_add writes 16 floats
    x = torch.ones(16, device="cuda")
    y = torch.ones(16, device="cuda")
    victim = torch.zeros(256, device="cuda")
    off = -64 if side == "end" else victim.numel() * 4
    _lib.check(pg.pg_add(victim.data_ptr() + off, x.data_ptr(), y.data_ptr(), 16, st), "pg_add")
    torch.cuda.synchronize()
    n2, rep = guard.check_all()
    print("after a deliberate 64-byte stray write:", n2 - n1, rep.strip()[-300:])
    assert n2 - n1 == 1
    del t, a, b, c, d
    lib.pg_guard_settle_retries.restype = ctypes.c_long
    print("live allocations:", lib.pg_guard_live(), " fills of fresh memory that did not stick:", lib.pg_guard_settle_retries())
    if len(sys.argv) > 1 and sys.argv[1] == "fault":
        # read 4 KB past the guarded edge: must fault (the process dies with the HSA memory-fault message)
        big = torch.zeros(1024, device="cuda")
        off = big.numel() * 4 + 4096 if side == "end" else -8192
        _lib.check(pg.pg_add(x.data_ptr(), big.data_ptr() + off, y.data_ptr(), 16, st), "pg_add")
        torch.cuda.synchronize()
        print("NO FAULT on an out-of-mapping read")
        return
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "fault"], capture_output=True, text=True, timeout=300)
    print("child (stray read past the guard): rc", p.returncode, "|", (p.stderr or "")[-300:].replace("\n", " | "))
    assert p.returncode != 0 and "NO FAULT" not in p.stdout
    print("guard selftest OK")


if __name__ == "__main__":
    os.environ["PG_GUARD"] = "1"
    main()
