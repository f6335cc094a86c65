"""Exhaustive bank-conflict check of the LDS layouts of conv_wino.hip (run on any machine: pure arithmetic).

ds_read_b128 is serviced in 4 groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), bank = (byte/4) mod 64
(MI355X_MICROARCH.md, LDS table): a group is conflict-free iff its 16 lanes hit 16 distinct 16-byte bank groups.

raw tile   granule = pix' * 4 + (quad ^ ((py >> 1) & 3)),  pix' = py * 18 + (px & 1) * 9 + (px >> 1)
           reads: lane = (tile ty 0..3, tx 0..7), pixel (2 ty + r, 2 tx + s), r in the wave's two rows, s 0..3, quad 0..3
exchange   dword = (pos * 32 + tile) * S + channel, S = 36: ds_write_b128 in 8 groups of 8 contiguous lanes over 32 banks
"""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]


def raw_worst(pitch=18):
    worst = 0
    for r in range(4):
        for s in range(4):
            for q in range(4):
                for grp in GROUPS:
                    banks = {}
                    for m in grp:
                        ty, tx = m >> 3, m & 7
                        py, px = 2 * ty + r, 2 * tx + s
                        g = (py * pitch + (px & 1) * (pitch // 2) + (px >> 1)) * 4 + (q ^ ((py >> 1) & 3))
                        banks.setdefault(g % 16, set()).add(g)
                    worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def exchange_write_worst(S=36):
    worst = 0
    for rr in range(4):
        for g0 in range(0, 64, 8):
            banks = {}
            for lane in range(g0, g0 + 8):
                d = (lane & 31) * S + 8 * rr + 4 * (lane >> 5)
                for k in range(4):
                    banks.setdefault((d + k) % 32, set()).add(d + k)
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst


if __name__ == "__main__":
    print("raw tile reads: worst", raw_worst(), "-way;  exchange writes: worst", exchange_write_worst(), "-way")
    assert raw_worst() == 1 and exchange_write_worst() == 1
