"""CPU model of tools/probe/g256p_probe.hip's index arithmetic (no GPU needed).

(1) DMA source offsets -> lane-linear LDS image of a unit ([256 rows][64 B], 16-byte chunk c of row r at position c ^ f4((r >> 2) & 3))
    -> ds_read_b128 fragment addresses: every element a lane feeds to the MFMA is the intended A(m, k) / B(n, k);
(2) LDS bank conflicts of the fragment reads under the guide's lane groups for ds_read_b128;
(3) the way out: accumulator block (mb, nb), lane (l15, lg), register r = C(row mb*16 + l15, column nb*16 + 4*lg + r) (operands swapped as
    in csrc/gemm256.hip); two adjacent column blocks are merged by v_permlane16_swap (odd 16-lane rows of the first register <-> even
    rows of the second) into 16-byte pieces; piece (p, mb) of a lane goes to byte offset c_lane + p*64 + mb*16*ldc*2 -- every element of
    the wave's 128 x 128 quarter must be written exactly once, to its own address."""
import numpy as np

UNIT = 16384


def f4(i):
    """chunk rotation key of a row: i = (row >> 2) & 3 -> 0, 2, 3, 1 (an XOR key per 4-row group chosen so that the four lanes of a
    hardware lane group of ds_read_b128 that share (row & 3) land on four different 16-byte bank groups)"""
    return (((i ^ (i >> 1)) & 1) << 1) | (i >> 1)


def fill_unit(rows_src):
    """rows_src(r) -> logical row of unit row r; image element = row * 32 + k (k = 0..31 of the K-step)."""
    img = -np.ones(UNIT // 2, dtype=np.int64)
    for wave in range(4):
        for j in range(4):
            for lane in range(64):
                r = (wave * 4 + j) * 16 + (lane >> 2)                 # row of the unit this lane's 16 bytes belong to
                chunk = (lane & 3) ^ f4((lane >> 4) & 3)            # source chunk: slot ^ f4((r >> 2) & 3)
                assert ((lane >> 4) & 3) == ((r >> 2) & 3)
                dst = (wave * 4 + j) * 1024 + lane * 16                # LDS-DMA writes lane-linearly
                for e in range(8):
                    img[dst // 2 + e] = rows_src(r) * 32 + chunk * 8 + e
    assert (img >= 0).all()
    return img


def banks_b128(addrs):
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    worst = 1
    for g in groups:
        cnt = {}
        for l in g:
            for d in range(4):
                cnt.setdefault(((addrs[l] // 4) + d) % 64, set()).add(addrs[l] // 4 + d)
        worst = max(worst, max(len(v) for v in cnt.values()))
    return worst


def check_fragments():
    img = fill_unit(lambda r: r)
    worst = 1
    for w in range(4):
        for half, base_row in (("A", (w >> 1) * 128), ("B", (w & 1) * 128)):
            for blk in range(8):
                addrs = []
                for lane in range(64):
                    l15, lg = lane & 15, lane >> 4
                    a = (base_row + l15) * 64 + ((lg ^ f4((l15 >> 2) & 3)) << 4) + blk * 1024
                    addrs.append(a)
                    for e in range(8):
                        assert img[a // 2 + e] == (base_row + blk * 16 + l15) * 32 + lg * 8 + e, (w, half, blk, lane, e)
                worst = max(worst, banks_b128(addrs))
    return worst


def permlane16_swap(vdst, src):
    """odd 16-lane rows of vdst <-> even rows of src (v_permlane16_swap_b32)"""
    v, s = list(vdst), list(src)
    for row in (0, 2):
        for i in range(16):
            v[(row + 1) * 16 + i], s[row * 16 + i] = src[row * 16 + i], vdst[(row + 1) * 16 + i]
    return v, s


def check_way_out(ldc=768):
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        written = {}
        for p in range(4):
            for mb in range(8):
                # accumulator blocks x = (mb, 2p), y = (mb, 2p + 1) as packed bf16 pairs: dword d of a lane = columns 4 lg + 2 d, + 1
                def dword(nb, d):
                    return [((mb * 16 + (lane & 15)), nb * 16 + 4 * (lane >> 4) + 2 * d) for lane in range(64)]      # (row, first col)
                x0, x1, y0, y1 = dword(2 * p, 0), dword(2 * p, 1), dword(2 * p + 1, 0), dword(2 * p + 1, 1)
                r0v, r0s = permlane16_swap(x0, y0)
                r1v, r1s = permlane16_swap(x1, y1)
                for lane in range(64):
                    piece = [r0v[lane], r1v[lane], r0s[lane], r1s[lane]]            # p.u[0..3]
                    l15, lg = lane & 15, lane >> 4
                    c_lane = ((wr * 128 + l15) * ldc + wc * 128 + (lg & 1) * 16 + (lg >> 1) * 8) * 2
                    off = c_lane + p * 64 + mb * 16 * ldc * 2
                    for d, (row, col) in enumerate(piece):
                        for h in range(2):
                            addr = off + d * 4 + h * 2
                            want = ((wr * 128 + row) * ldc + wc * 128 + col + h) * 2
                            assert addr == want, (wave, p, mb, lane, d, h, addr, want)
                            assert addr not in written
                            written[addr] = True
        assert len(written) == 128 * 128, len(written)
    return True


if __name__ == "__main__":
    print("fragment reads: worst bank conflict", check_fragments(), "-way")
    print("way out (permlane16_swap pieces -> C):", "every element once, at its address" if check_way_out() else "WRONG")
    print("g256p layout model OK")
