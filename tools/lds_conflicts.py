#!/usr/bin/env python3
"""Bank-conflict census of the igemm patch layout (eae_igemm.hip.h: swz_col / Geo::frag_lane) for every tile geometry the launcher
uses: for each ds_read_b128 fragment read (m-tile, patch offset) the 64 lane addresses are grouped as the hardware does
(MI355X_MICROARCH.md, LDS: 4 groups of 16 lanes, bank = (addr/4) % 64, 16-byte slots) and the number of LDS cycles is counted
(distinct addresses per 16-byte slot, max over slots, summed over the groups; 4 = conflict-free).  Diagnostic tool."""
import itertools
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def cycles(addrs):
    """LDS cycles of one ds_read_b128 wave-instruction given the 64 byte addresses"""
    tot = 0
    for g in GROUPS:
        slots = {}
        for l in g:
            a = addrs[l]
            slots.setdefault((a // 16) % 16, set()).add(a // 16)
        tot += max(len(v) for v in slots.values())
    return tot


def swz_col(col, kg, row=0, mode=0):
    """swz_px<SWZ> of eae_igemm.hip.h (mode = swizzle class)"""
    r, j = col >> 2, col & 3
    if mode == 0:
        jj, cc = (j ^ r) & 3, (kg ^ r) & 3
    elif mode == 1:
        jj, cc = (j ^ (row >> 1)) & 3, (kg ^ (2 * r)) & 3
    elif mode == 2:
        jj, cc = j, (kg ^ (2 * r)) & 3
    elif mode == 3:
        jj, cc = j, (kg ^ (2 * row)) & 3
    else:
        raise ValueError(mode)
    return ((col & ~3) | jj) * 32 + cc * 8


def swz_class(kind, TW):
    """Geo::SWZ"""
    return (0 if TW == 16 else 1) if kind == 0 else (2 if TW == 16 else 3)


def geo(kind, TW, TH, NI):
    P = NI * TH * TW
    PH = 2 * TH + 1 if kind == 0 else TH + 1
    PW = 2 * TW + 1 if kind == 0 else TW + 1
    PWS = (PW + 3) & ~3
    return dict(P=P, PH=PH, PW=PW, PWS=PWS, RS=PWS * 32, MUL=2 if kind == 0 else 1, NKX=3 if kind == 0 else 2)


def census(kind, TW, TH, NI, mode, rs_pad=0):
    g = geo(kind, TW, TH, NI)
    RS = g["RS"] + rs_pad
    noff = 9 if kind == 0 else 4
    worst, total, n = 0, 0, 0
    for mt in range(g["P"] // 16):
        for o in range(noff):
            orow, ocol = (o // 3, o % 3) if kind == 0 else (o >> 1, o & 1)
            addrs = []
            for lane in range(64):
                i, kg = lane & 15, lane >> 4
                pos = mt * 16 + i
                img, ty, tx = pos // (TH * TW), (pos // TW) % TH, pos % TW
                row = g["MUL"] * ty + orow
                col = g["MUL"] * tx + ocol
                el = (img * g["PH"] + row) * RS + swz_col(col, kg, row, mode)
                addrs.append(el * 2)
            c = cycles(addrs)
            worst = max(worst, c); total += c; n += 1
    return worst, total / n


if __name__ == "__main__":
    geos = [("conv 16x8x1", 0, 16, 8, 1), ("conv 8x8x2", 0, 8, 8, 2), ("conv 4x4x8", 0, 4, 4, 8), ("conv 8x8x1", 0, 8, 8, 1), ("conv 4x4x4", 0, 4, 4, 4),
            ("deconv 16x8x1", 1, 16, 8, 1), ("deconv 8x8x1", 1, 8, 8, 1), ("deconv 4x4x4", 1, 4, 4, 4), ("deconv 16x4x1", 1, 16, 4, 1)]
    bad = 0
    for name, kind, TW, TH, NI in geos:
        w0, a0 = census(kind, TW, TH, NI, 0)
        w, a = census(kind, TW, TH, NI, swz_class(kind, TW))
        bad += w != 4
        print(f"{name:16s} class {swz_class(kind, TW)}: worst {w:2d} avg {a:5.2f} LDS cycles per fragment read   (round-2 layout: worst {w0:2d} avg {a0:5.2f})")
    sys.exit(1 if bad else 0)
