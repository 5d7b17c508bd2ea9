#!/usr/bin/env python
"""CPU checks for the patch-slab convolution (open-muse_amd/csrc/conv_dma.hip, namespace cslab).

1. bank model: ds_read_b128 is serviced in four 16-lane groups (MI355X_MICROARCH.md, LDS table); a 64-byte-row image read at
   16 consecutive rows starting ANYWHERE must put each group's 16 chunks on 16 distinct 16-byte bank slots.  Enumerates the
   swizzles pos = g ^ f((row >> 2) & 3) and prints the LDS cycles per start row (4 = conflict-free).
2. address emulation: replays the kernel's DMA placement (piece / lane -> LDS byte) and its fragment read addresses
   (baseA[e][b] + immediate) and checks every lane of every (wave row, fragment, tap) reads the pixel / channel chunk the
   implicit GEMM needs, zero padding included."""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles(base, f):
    tot = 0
    for G in GROUPS:
        cnt = {}
        for l in G:
            pr, g = l & 15, l >> 4
            row = base + pr
            slot = (row * 4 + (g ^ f[(row >> 2) & 3])) % 16
            cnt[slot] = cnt.get(slot, 0) + 1
        tot += max(cnt.values())
    return tot


def search():
    res = []
    for f in itertools.product(range(4), repeat=4):
        c = [cycles(b, f) for b in range(16)]
        res.append((max(c), sum(c), f))
    res.sort()
    return res


PLANE, SLAB = 21 * 1024, 42 * 1024


def emulate(H, W, img, ty, tx):
    y0, x0 = ty * 16, tx * 16
    lds = {}
    for wave in range(8):
        for k in range(3):
            piece = wave + 8 * k
            if piece > 20:
                piece -= 8
            for lane in range(64):
                srcchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1)
                r = piece * 16 + (lane >> 2)
                sy, sx = divmod(r, 18)
                y, x = y0 - 1 + sy, x0 - 1 + sx
                ok = r < 324 and 0 <= y < H and 0 <= x < W
                val = (img, y, x, srcchunk) if ok else "zero"
                addr = piece * 1024 + lane * 16
                assert lds.get(addr, val) == val
                lds[addr] = val
    bad = 0
    for wm in range(4):
        for lane in range(64):
            pr, g = lane & 15, lane >> 4
            P = 72 * wm + pr
            base = {}
            for e in range(4):
                for b in range(2):
                    bit2 = ((P >> 2) & 1) ^ (((P & 3) + e) >> 2) ^ b
                    base[e, b] = (P << 6) | ((g << 4) ^ (bit2 << 5))
            for tap in range(9):
                ky, kx = divmod(tap, 3)
                for i in range(4):
                    cp = (i + ky) * 18 + kx
                    addr = base[cp & 3, (cp >> 2) & 1] + cp * 64
                    y, x = y0 + wm * 4 + i + ky - 1, x0 + pr + kx - 1
                    want = (img, y, x, g) if (0 <= y < H and 0 <= x < W) else "zero"
                    if lds.get(addr) != want:
                        bad += 1
    return bad


if __name__ == "__main__":
    r = search()
    print("best swizzles (max cycles, sum over 16 start rows, f):", r[:4])
    print("aligned-only swizzle {0,3,2,1}:", [cycles(b, (0, 3, 2, 1)) for b in range(8)])
    for H, W in ((16, 16), (32, 32), (64, 48)):
        for ty in range(H // 16):
            for tx in range(W // 16):
                assert emulate(H, W, 3, ty, tx) == 0, (H, W, ty, tx)
    print("address emulation: every fragment lane reads its pixel / chunk (incl. zero padding)")
