"""CPU emulation of the GLOBAL address generation of conv_wgrad_b3_kernel (x-copy) and conv_wgrad_b3s_kernel
(shifted dy), csrc/conv_wgrad_b3.hip: every float read must lie inside its tensor. Restates the host-side tile
selection and the per-slot offsets / guards of PG_WB_ISSUE / PG_WS_ISSUE.
usage: python tools/exp/emu_wgrad_bounds.py"""
import itertools

WB_THREADS, WB_CI, WB_DS, WB_XS, BUDGET = 256, 32, 3, 3, 76 * 1024


def check(name, idx, numel, what):
    if idx < 0 or idx >= numel:
        raise AssertionError(f"{name}: {what} reads element {idx} of {numel}")


def old_kernel(N, Cin, Cout, H, W, taps):
    """taps: list of (dr, dc). Returns None if the host declines, else the number of checked reads."""
    T = len(taps)
    if W % 4 or Cout % 32 or Cin % WB_CI:
        return None
    MR = 2 if Cout % 64 == 0 else 1
    if MR == 1 and T == 1:
        return None
    if T not in (1, 2, 3, 4, 6, 9):
        return None
    dcs = []
    for dr, dc in taps:
        if dc < -1 or dc > 1:
            return None
        if dc not in dcs:
            dcs.append(dc)
    min_dr = min(t[0] for t in taps); max_dr = max(t[0] for t in taps)
    hr = max_dr - min_dr
    PBR = (W + 7) // 8

    def pick_rows(co_, ci_, dcap, xcap=None):
        xcap = xcap or dcap
        best = 0
        for tr in range(1, H + 4):
            if (tr * PBR) % 4:
                continue
            dslots, xslots = co_ * tr * PBR, ci_ * (tr + hr) * PBR
            if dslots > dcap or xslots > xcap or (3 * dslots + 3 * len(dcs) * xslots) * 16 > BUDGET:
                break
            best = tr
            if tr >= H:
                break
        return best

    # round 5: big tiles (pg_wgrad_b3_launch): 64 x channels per workgroup for <= 2 taps, 128 dy channels for one tap
    big = MR == 2 and T <= 2 and Cin % 64 == 0
    TR = 0
    if big and T == 1 and Cout % 128 == 0:
        TR = pick_rows(128, 64, 1024, 512)
        if TR > 0:
            MR = 4
    if big and TR == 0:
        TR = pick_rows(64, 64, 1024)
        if TR == 0:
            big = False
    wb_ci = 64 if big else WB_CI
    wb_co = 32 * MR
    waves = 8 if big else (8 if (MR == 2 and T <= 4) else 4)
    if not big:
        TR = pick_rows(wb_co, wb_ci, 1024 if waves == 8 else WB_DS * WB_THREADS)
    if TR == 0:
        return None
    xh = TR + hr
    tiles_per_img = (H + TR - 1) // TR
    want_m1, want_p1 = -1 in dcs, 1 in dcs
    wpart = (W & 7) != 0
    nx, ndy, reads = N * Cin * H * W, N * Cout * H * W, 0
    for n in range(N):
        for tile in range(tiles_per_img):
            row0 = tile * TR
            for co0 in range(0, Cout, wb_co):
                dyb = ((n * Cout + co0) * H + row0) * W
                for e in range(wb_co * TR * PBR):
                    i = e & 15; e2 = e >> 4
                    cb = e2 % PBR; e2 //= PBR
                    tr = e2 % TR; cot = e2 // TR
                    goff = ((cot * 16 + i) * H + tr) * W + 8 * cb
                    if row0 + tr < H:
                        half = wpart and cb == PBR - 1
                        for f in range(4):
                            check("old", dyb + goff + f, ndy, "dy p[0]")
                        for f in range(4):
                            check("old", dyb + goff + (0 if half else 4) + f, ndy, "dy p[1]")
                            reads += 8
            for ci0 in range(0, Cin, wb_ci):
                xb = ((n * Cin + ci0) * H + (row0 + min_dr)) * W
                for e in range(wb_ci * xh * PBR):
                    i = e & 15; e2 = e >> 4
                    cb = e2 % PBR; e2 //= PBR
                    tr = e2 % xh; cit = e2 // xh
                    goff = ((cit * 16 + i) * H + tr) * W + 8 * cb
                    ir = row0 + min_dr + tr
                    if 0 <= ir < H:
                        half = wpart and cb == PBR - 1
                        q = xb + goff
                        for f in range(4):
                            check("old", q + f, nx, "x p[0]")
                            check("old", q + (0 if half else 4) + f, nx, "x p[1]")
                        if want_m1:
                            check("old", q + (0 if cb == 0 else -1), nx, "x prev")
                        if want_p1:
                            check("old", q + (3 if cb == PBR - 1 else 8), nx, "x next")
                        reads += 10
    return reads


def b3s_kernel(N, Cin, Cout, H, W, taps):
    T = len(taps)
    if W % 4 or Cout % 32 or Cin % WB_CI or T < 2 or T > 9:
        return None
    min_dr = min(t[0] for t in taps); max_dr = max(t[0] for t in taps)
    min_dc = min(t[1] for t in taps); max_dc = max(t[1] for t in taps)
    NR, NC = max_dr - min_dr + 1, max_dc - min_dc + 1
    if min_dc < -1 or max_dc > 1 or NR > 3 or NR * NC != T or (NR, NC) not in ((3, 3), (2, 2), (1, 3), (2, 1), (2, 3)):
        return None
    hr = NR - 1
    PBR = (W + 7) // 8
    TR = 0
    for tr in range(1, H + 4):
        if (tr * PBR) % 4:
            continue
        dslots, xslots = 32 * tr * PBR, WB_CI * (tr + hr) * PBR
        if dslots > WB_DS * WB_THREADS or xslots > WB_XS * WB_THREADS or (3 * NC * dslots + 3 * xslots) * 16 > BUDGET:
            break
        TR = tr
        if tr >= H:
            break
    if TR == 0:
        return None
    xh = TR + hr
    tiles_per_img = (H + TR - 1) // TR
    want_prev, want_next = min_dc + NC - 1 >= 1, min_dc <= -1
    wpart = (W & 7) != 0
    nx, ndy, reads = N * Cin * H * W, N * Cout * H * W, 0
    for n in range(N):
        for tile in range(tiles_per_img):
            row0 = tile * TR
            for co0 in range(0, Cout, 32):
                dyb = ((n * Cout + co0) * H + row0) * W
                for e in range(32 * TR * PBR):
                    i = e & 15; e2 = e >> 4
                    cb = e2 % PBR; e2 //= PBR
                    tr = e2 % TR; cot = e2 // TR
                    goff = ((cot * 16 + i) * H + tr) * W + 8 * cb
                    if row0 + tr < H:
                        half = wpart and cb == PBR - 1
                        q = dyb + goff
                        for f in range(4):
                            check("b3s", q + f, ndy, "dy p[0]")
                            check("b3s", q + (0 if half else 4) + f, ndy, "dy p[1]")
                        if want_prev:
                            check("b3s", q + (0 if cb == 0 else -1), ndy, "dy prev")
                        if want_next:
                            check("b3s", q + (3 if cb == PBR - 1 else 8), ndy, "dy next")
                        reads += 10
            for ci0 in range(0, Cin, WB_CI):
                xb = ((n * Cin + ci0) * H + (row0 + min_dr)) * W
                for e in range(WB_CI * xh * PBR):
                    i = e & 15; e2 = e >> 4
                    cb = e2 % PBR; e2 //= PBR
                    tr = e2 % xh; cit = e2 // xh
                    goff = ((cit * 16 + i) * H + tr) * W + 8 * cb
                    ir = row0 + min_dr + tr
                    if 0 <= ir < H:
                        half = wpart and cb == PBR - 1
                        for f in range(4):
                            check("b3s", xb + goff + f, nx, "x p[0]")
                            check("b3s", xb + goff + (0 if half else 4) + f, nx, "x p[1]")
                        reads += 8
    return reads


def taps_of(kh, kw, ph, pw):
    return [(u - ph, v - pw) for u in range(kh) for v in range(kw)]


if __name__ == "__main__":
    shapes = []
    for (kh, kw, ph, pw) in ((1, 1, 0, 0), (3, 3, 1, 1), (2, 2, 1, 1), (1, 3, 0, 1), (2, 1, 2, 0), (2, 3, 1, 1), (1, 2, 0, 1)):
        for (h, w) in ((28, 28), (12, 12), (10, 20), (32, 36), (7, 24), (16, 16), (64, 64), (9, 16), (8, 8), (18, 28)):
            for (cin, cout) in ((32, 32), (32, 64), (64, 64), (128, 256), (96, 96), (64, 128), (128, 128), (256, 256), (192, 128), (320, 64)):
                shapes.append((1, cin, cout, h, w, taps_of(kh, kw, ph, pw)))
                shapes.append((2, cin, cout, h, w, taps_of(kh, kw, ph, pw)))
    n_old = n_s = 0
    for (n, cin, cout, h, w, taps) in shapes:
        r = old_kernel(n, cin, cout, h, w, taps)
        n_old += r is not None
        r = b3s_kernel(n, cin, cout, h, w, taps)
        n_s += r is not None
    print(f"checked {len(shapes)} shapes: x-copy kernel took {n_old}, shifted-dy kernel took {n_s}: all reads in bounds")
