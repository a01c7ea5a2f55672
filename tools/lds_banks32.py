#!/usr/bin/env python
"""Bank-conflict search for the 32x32x16-MFMA fragment reads of the halo-resident 3x3 convolution.

A lane reads, for pixel tile mt (32 pixels) and K half kh, the 16-byte chunk 2*kh + (lane >> 5) of halo pixel
(lane & 31) of the tile; LDS rows are 64 B (one halo pixel x 32 channels), chunk' = chunk ^ g(halo row, halo column).
gfx950 services ds_read_b128 in 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): inside a group the 16
chunks must fall on 16 distinct 16-byte bank groups (address / 16 mod 16).  4.0 cycles per read = conflict free.
"""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[lane + 32 for lane in g] for g in GROUPS]


def cycles(addr16):
    total = 0
    for g in GROUPS:
        banks = {}
        for lane in g:
            banks.setdefault(addr16[lane] % 16, set()).add(addr16[lane])
        total += max(len(v) for v in banks.values())
    return total


def reads(W, TH, TI, BM):
    out = []
    for wm in range(BM // 64):
        for mt in range(2):
            for kh in range(3):
                for kw in range(3):
                    lanes = []
                    for lane in range(64):
                        m = wm * 64 + mt * 32 + (lane & 31)
                        ti, ty, tx = m // W // TH, m // W % TH, m % W
                        lanes.append((ti * (TH + 2) + ty + kh, tx + kw))
                    out.append(lanes)
    return out


def score(W, TH, TI, BM, g):
    stride = W + 2
    R = reads(W, TH, TI, BM)
    tot = 0
    for lanes in R:
        for khalf in range(2):
            tot += cycles([(hr * stride + hx) * 4 + ((2 * khalf + (lane >> 5)) ^ g(hr, hx)) for lane, (hr, hx) in enumerate(lanes)])
    return tot / (2 * len(R))


CANDS = {}
for a, s in itertools.product(range(4), range(4)):
    CANDS['(%d*hr + (hx>>%d))&3' % (a, s)] = (lambda a, s: lambda hr, hx: (a * hr + (hx >> s)) & 3)(a, s)
for a, s in itertools.product(range(1, 4), range(1, 3)):
    CANDS['((hr>>1)*%d + (hx>>%d))&3' % (a, s)] = (lambda a, s: lambda hr, hx: ((hr >> 1) * a + (hx >> s)) & 3)(a, s)

if __name__ == '__main__':
    shapes = ((32, 8, 1, 256), (32, 4, 1, 128), (16, 8, 1, 128), (16, 16, 1, 256), (8, 8, 2, 128), (8, 8, 4, 256), (4, 4, 8, 128),
              (64, 4, 1, 256), (64, 2, 1, 128), (128, 1, 1, 128), (128, 2, 1, 256))
    for sh in shapes:
        res = sorted((score(*sh, g), n) for n, g in CANDS.items())
        print(sh, ' best:', ', '.join('%s %.2f' % (n, v) for v, n in res[:4]))
    # one formula for everything?
    tot = {n: sum(score(*sh, g) for sh in shapes) / len(shapes) for n, g in CANDS.items()}
    print('overall best:', sorted((v, n) for n, v in tot.items())[:5])

    # W = 4 (layer 4: 8 images x 4 rows per 128-pixel tile): wider family
    best = []
    for a, b, c, d, e in itertools.product(range(4), repeat=5):
        g = (lambda a, b, c, d, e: lambda hr, hx: (a * hr + b * hx + c * (hr >> 1) + d * (hx >> 1) + e * (hr >> 2)) & 3)(a, b, c, d, e)
        best.append((score(4, 4, 8, 128, g), (a, b, c, d, e)))
    best.sort()
    print('W=4 family (a*hr + b*hx + c*(hr>>1) + d*(hx>>1) + e*(hr>>2)) & 3:', best[:6])
