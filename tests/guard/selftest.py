"""Self-test of the guard allocator on the GPU (python tests/guard/selftest.py): copies in and out of guarded tensors
are intact, clean tensors raise no violation, fresh tensors are poisoned, and deliberate 64-byte writes in front of and
behind a tensor are each reported with the right offsets. (PG_GUARD_MODE=vmm additionally tries the guard-page mode:
a read past the guarded edge must fault in a child process.)"""

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

    vmm = os.environ.get("PG_GUARD_MODE") == "vmm"
    lib = guard.install()
    from pytorch_generative_amd import _lib

    pg = _lib.load()
    st = ctypes.c_void_p(0)
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
    a = torch.ones(6, device="cuda")
    b = torch.randn(1000, device="cuda") * 2
    c = torch.zeros((), device="cuda", dtype=torch.int64)
    d = (a + 1).sum()
    torch.cuda.synchronize()
    n1, rep = guard.check_all()
    print("after a few ATen ops:", n1, rep.strip()[-400:])
    assert n1 == 0, "violations without any out-of-bounds access: the harness itself is wrong"
    # deliberate damage: pg_add writes 16 floats = 64 bytes just in front of / just behind a 1 KB tensor
    x = torch.ones(16, device="cuda")
    y = torch.ones(16, device="cuda")
    victim = torch.zeros(256, device="cuda")
    _lib.check(pg.pg_add(x.data_ptr(), y.data_ptr(), victim.data_ptr() - 64, 16, st), "pg_add")  # (a, b, out)
    torch.cuda.synchronize()
    n2, rep = guard.check_all()
    print("after a 64-byte stray write in front:", n2 - n1, rep.strip().splitlines()[-1][-260:] if rep.strip() else "")
    assert n2 - n1 == 1 and "offsets -64..-1" in rep
    if not vmm:
        _lib.check(pg.pg_add(x.data_ptr(), y.data_ptr(), victim.data_ptr() + 1024, 16, st), "pg_add")
        torch.cuda.synchronize()
        n3, rep = guard.check_all()
        print("after a 64-byte stray write behind:", n3 - n2, rep.strip().splitlines()[-1][-260:])
        assert n3 - n2 == 1 and "offsets 1024..1087" in rep
        assert float(victim.abs().sum()) == 0.0
    del t, a, b, c, d
    print("live allocations:", lib.pg_guard_live())
    if vmm:
        if len(sys.argv) > 1 and sys.argv[1] == "fault":
            bigt = torch.zeros(1024, device="cuda")
            _lib.check(pg.pg_add(bigt.data_ptr() + 4096 + 4096, y.data_ptr(), x.data_ptr(), 16, st), "pg_add")
            torch.cuda.synchronize()
            print("NO FAULT on an out-of-mapping read")
            return
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "fault"], capture_output=True, text=True, timeout=300)
        print("child (stray read past the guard page): rc", p.returncode, "|", (p.stderr or "")[-300:].replace("\n", " | "))
    print("guard selftest OK")


if __name__ == "__main__":
    os.environ["PG_GUARD"] = "1"
    main()
