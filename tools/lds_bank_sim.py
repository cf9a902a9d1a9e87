#!/usr/bin/env python3
"""LDS bank-conflict model for the operand reads of k_jac_fused's matrix phase (CPU only).

gfx950 LDS: 64 banks of 4 bytes for 8- and 16-byte reads; a wave64 ds_read_b64 is served in two groups of
32 lanes, a ds_read_b128 in four groups of 16 lanes (the lane sets below); lanes of a group that hit the same
bank with different addresses cost one extra cycle each.  The script prints, for a column stride RS (doubles)
and each wavefront of a workgroup, the LDS cycles of the four reads of one group of eight rows:
N (column c), N1 (column c+4 mod 16), AR / BR (the half-and-half reads of columns c, c+-8).  Conflict-free is
(4, 4, 2, 2); the library's geometry (RS = 4 mod 8, wavefront buffers a multiple of 32 doubles apart) gives
that for every wavefront, the earlier stride (RS = 2, 14, 18, 30 mod 32) does not.
    python tools/lds_bank_sim.py [points_per_view]
"""
import sys

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G64 = [list(range(32)), list(range(32, 64))]


def cycles(addr_dw, width_dw, groups):
    total = 0
    for g in groups:
        banks = {}
        for lane in g:
            for k in range(width_dw):
                a = addr_dw[lane] + k
                banks.setdefault(a % 64, set()).add(a)
        total += max(len(v) for v in banks.values())
    return total


def reads(RS, wave, pad_waves, icx=2, icy=3, cbase=82, npriv=14):
    stride = (npriv * RS + 31) // 32 * 32 if pad_waves else npriv * RS
    jt = cbase + 2 * RS + wave * stride

    def colbase(c):
        if c == icx:
            return cbase
        if c == icy:
            return cbase + RS
        return jt + (c - 2 if c > icy else c) * RS

    N, N1, A, B = [0] * 64, [0] * 64, [0] * 64, [0] * 64
    for lane in range(64):
        c, rq = lane & 15, lane >> 4
        N[lane] = 2 * (colbase(c) + 2 * rq)
        N1[lane] = 2 * (colbase((c + 4) & 15) + 2 * rq)
        A[lane] = 2 * (colbase(c) + 2 * rq) if c < 8 else 2 * (colbase(c - 8) + 2 * rq + 1)
        B[lane] = 2 * (colbase(c + 8) + 2 * rq) if c < 8 else 2 * (colbase(c) + 2 * rq + 1)
    return cycles(N, 4, G128), cycles(N1, 4, G128), cycles(A, 2, G64), cycles(B, 2, G64)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    CH = 64 if P >= 64 else (P + 7) // 8 * 8
    old = 2 * CH
    while old % 32 not in (2, 14, 18, 30):
        old += 2
    for name, RS, pad in (("library geometry", 2 * CH + 4, True), ("earlier stride", old, False)):
        print(f"{name}: RS = {RS} doubles")
        for w in range(4):
            print(f"  wavefront {w}: LDS cycles (N, N1, AR, BR) = {reads(RS, w, pad)}")


if __name__ == "__main__":
    main()
