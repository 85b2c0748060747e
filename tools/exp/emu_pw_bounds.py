"""CPU emulation of the global addresses of conv_b3_pw_kernel (csrc/conv_b3_kernels.h): B-fragment loads of x, the
epilogue's operand loads and stores. Every access must lie inside its tensor.
usage: python tools/exp/emu_pw_bounds.py"""


def plan(Kc, M):
    if Kc % 8 or Kc < 16 or M < 16:
        return None
    MT = 4 if M >= 64 else (M + 15) // 16
    for CIB in (32, 16, 8):
        if Kc % CIB:
            continue
        return CIB, CIB // 8, MT
    return None


def pw(N, Cin, H, W, Cout):
    L = H * W
    pl = plan(Cin, Cout)
    if pl is None or L < 256 or W > 256 or L % 2:
        return None
    CIB, cgs, MT = pl
    nchunk = Cin // CIB
    if nchunk * MT * 3 * 1024 > 24 * 1024:
        return None
    tpi = (L + 31) // 32
    nitems = N * tpi
    nin, nout = N * Cin * L, N * Cout * L
    chunks_y = (Cout + 63) // 64
    for by in range(chunks_y):
        co0 = by * 64
        for it in range(nitems):
            ni, t0 = it // tpi, (it % tpi) * 32
            for lane in range(64):
                kq, jc = lane >> 4, lane & 15
                kact = kq < cgs
                lane_in = (8 * kq if kact else 0) * L + 2 * jc
                for j in range(nchunk):
                    ok = kact and t0 + 2 * jc < L
                    lo = lane_in if ok else 0
                    sb = (ni * Cin + j * CIB) * L + t0
                    for c in range(8):
                        a = sb + c * L + lo
                        assert 0 <= a and a + 1 < nin, ("x load", N, Cin, H, W, Cout, it, lane, j, c, a, nin)
                half, px = lane >> 5, lane & 31
                sok = t0 + px < L
                so = (ni * Cout + co0) * L + t0
                cvalid = Cout - co0 - 8 * half
                lo = (8 * half * L + px) if (sok and cvalid > 0) else 0
                for m in range(MT):
                    for c in range(8):
                        cc = m * 16 + c
                        a = so + (cc if cc < cvalid else 0) * L + lo       # operand load (always issued)
                        assert 0 <= a < nout, ("operand", N, Cin, H, W, Cout, it, lane, m, c, a, nout, cvalid)
                        if sok and cc < cvalid:
                            a = so + cc * L + lo
                            assert 0 <= a < nout, ("store", N, Cin, H, W, Cout, it, lane, m, c, a, nout)
    return nitems


if __name__ == "__main__":
    took = 0
    for N in (1, 2):
        for (h, w) in ((28, 28), (16, 16), (36, 36), (10, 64), (16, 28), (28, 16), (18, 16)):
            for cin in (16, 24, 32, 40, 48, 64):
                for cout in (16, 24, 32, 36, 48, 64, 66, 68, 72, 96, 128, 136):
                    took += pw(N, cin, h, w, cout) is not None
    print(f"barrier-free 1x1 kernel: {took} shapes taken, every load / operand / store address in bounds")
