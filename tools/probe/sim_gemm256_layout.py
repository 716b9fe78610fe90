"""CPU model of gemm256.hip's index arithmetic (no GPU needed): DMA source -> lane-linear LDS image -> fragment reads.
Checks (1) every fragment element a lane feeds to the MFMA is the intended A(m,k) / B(n,k), for the k-contiguous
([128 rows][64 k], ds_read_b128) and the reduction-slow ([64 k][128 cols], ds_read_b64_tr_b16) unit images, and
(2) LDS bank conflicts of the read patterns under the guide's lane grouping."""
import numpy as np

UNIT = 16384


def fill_kc(unit_rows_src):
    """unit_rows_src(ur) -> logical row id. Returns LDS image as array [UNIT//2] of (row, k) tuples encoded row*64+k."""
    img = -np.ones(UNIT // 2, dtype=np.int64)
    for wave in range(8):
        for j in range(2):
            for lane in range(64):
                ur = wave * 16 + j * 8 + (lane >> 3)
                kc = (lane & 7) ^ (j * 4 + (lane >> 4))
                dst = (wave * 2 + j) * 1024 + lane * 16
                for e in range(8):
                    img[dst // 2 + e] = unit_rows_src(ur) * 64 + kc * 8 + e
    return img


def fill_tr(unit_col_src):
    img = -np.ones(UNIT // 2, dtype=np.int64)
    for wave in range(8):
        for j in range(2):
            for lane in range(64):
                kr = wave * 8 + j * 4 + (lane >> 4)
                f8 = (lane >> 4) | ((wave & 1) << 2)
                c = ((((lane & 15) >> 1) ^ f8) << 4) + (lane & 1) * 8
                dst = (wave * 2 + j) * 1024 + lane * 16
                for e in range(8):
                    img[dst // 2 + e] = unit_col_src(c + e) * 64 + kr
    return img


def banks_ok_b128(addrs):
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    worst = 1
    for g in groups:
        cnt = {}
        for l in g:
            for d in range(4):
                b = ((addrs[l] // 4) + d) % 64
                cnt.setdefault(b, set()).add(addrs[l] // 4 + d)
        worst = max(worst, max(len(v) for v in cnt.values()))
    return worst


def banks_ok_b64(addrs):
    worst = 1
    for g in (range(0, 32), range(32, 64)):
        cnt = {}
        for l in g:
            for d in range(2):
                cnt.setdefault(((addrs[l] // 4) + d) % 64, set()).add(addrs[l] // 4 + d)
        worst = max(worst, max(len(v) for v in cnt.values()))
    return worst


def check_kc(isA):
    worst = 1
    for half in range(2):                      # U0/U3 (A) or U1/U2 (B)
        if isA:
            src = lambda ur: (ur >> 6) * 128 + (ur & 63) + 64 * half
        else:
            src = lambda ur: (ur >> 5) * 64 + (ur & 31) + 32 * half
        img = fill_kc(src)
        assert (img >= 0).all()
        for w in range(8):
            wr, wc = w >> 2, w & 3
            for blk in range(4 if isA else 2):
                for kk in range(2):
                    addrs = []
                    for lane in range(64):
                        l15, lg, sw = lane & 15, lane >> 4, (lane >> 1) & 7
                        base = ((wr * 64 if isA else wc * 32) + l15) * 128 + (((kk * 4 + lg) ^ sw) << 4)
                        a = base + blk * 2048
                        addrs.append(a)
                        row = (wr * 128 + half * 64 + blk * 16 + l15) if isA else (wc * 64 + half * 32 + blk * 16 + l15)
                        for e in range(8):
                            assert img[a // 2 + e] == row * 64 + kk * 32 + lg * 8 + e, (isA, half, w, blk, kk, lane, e)
                    worst = max(worst, banks_ok_b128(addrs))
    return worst


def check_tr(isA):
    worst = 1
    for half in range(2):
        if isA:
            src = lambda c: (c >> 6) * 128 + (c & 63) + 64 * half
        else:
            src = lambda c: (c >> 5) * 64 + (c & 31) + 32 * half
        img = fill_tr(src)
        assert (img >= 0).all()
        for w in range(8):
            wr, wc = w >> 2, w & 3
            for blk in range(4 if isA else 2):
                for kk in range(2):
                    for hi in range(2):
                        addrs = []
                        for lane in range(64):
                            l15, lg = lane & 15, lane >> 4
                            f8 = (l15 >> 2) | ((lg & 1) << 2)
                            krow = lg * 8 + (l15 >> 2)
                            u = (wr * 4 + blk) if isA else (wc * 2 + blk)
                            addrs.append(krow * 256 + (((u ^ f8) & 7) << 5) + (lane & 3) * 8 + kk * 8192 + hi * 1024)
                        worst = max(worst, banks_ok_b64(addrs))
                        # hardware transpose: lane t of a 16-lane group, element j <- element (t & 3) of the 8-byte chunk
                        # addressed by lane 4*j + (t >> 2) of the same group
                        for lane in range(64):
                            g0, t = lane & ~15, lane & 15
                            lg = lane >> 4
                            row = (wr * 128 + half * 64 + blk * 16 + t) if isA else (wc * 64 + half * 32 + blk * 16 + t)
                            for jj in range(4):
                                srcl = g0 + 4 * jj + (t >> 2)
                                got = img[addrs[srcl] // 2 + (t & 3)]
                                assert got == row * 64 + kk * 32 + lg * 8 + hi * 4 + jj, (isA, half, w, blk, kk, hi, lane, jj, got)
    return worst


if __name__ == "__main__":
    print("k-contiguous A: worst bank conflict", check_kc(True), "-way")
    print("k-contiguous B: worst bank conflict", check_kc(False), "-way")
    print("reduction-slow A: worst bank conflict", check_tr(True), "-way")
    print("reduction-slow B: worst bank conflict", check_tr(False), "-way")
    print("layout model OK")
