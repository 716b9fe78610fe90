"""Host simulation of the "Z image" of csrc/attention_pair.inc: [rows][128 B], 16-byte chunk c of row r stored at chunk position
c ^ zkey(r), zkey(r) = bit-reverse3((r >> 1) & 7).  Checks, for every lane / tile / step, that
  * the LDS-DMA placement + the per-lane read addresses of z_rows / z_cols return the logical element the un-permuted layout
    would (so the permutation is invisible to the MFMA fragments), and
  * the addresses one hardware pass serves together fall into distinct 16-byte bank groups (64 banks x 4 B = 16 groups):
    ds_read_b128 in its four non-contiguous 16-lane groups, ds_read_b64_tr_b16 in passes of 32 lanes.
Run: python tools/probe/sim_zimage_layout.py   (no GPU needed; tests/test_tools_cpu.py runs it too)."""
import numpy as np

ZP = 128


def _rng(*spans):
    return [l for a, b in spans for l in range(a, b + 1)]


# ds_read_b128 lane groups of gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, LDS table)
B128_GROUPS = [_rng((0, 3), (12, 15), (20, 27)), _rng((4, 11), (16, 19), (28, 31)), _rng((32, 35), (44, 47), (52, 59)), _rng((36, 43), (48, 51), (60, 63))]


def zkey(r):
    x = (r >> 1) & 7
    return ((x & 1) << 2) | (x & 2) | (x >> 2)


def build_image(mat):
    """mat: [rows, 64] uint16 -> LDS bytes as the DMA writes them (lane-linear 16-byte pieces, permuted source chunk)."""
    rows = mat.shape[0]
    lds = np.zeros(((rows * ZP + 1023) // 1024 * 1024) // 2, dtype=np.uint16)
    for g in range(lds.size // 8):
        row, pos = g >> 3, g & 7
        c = pos ^ zkey(row)
        if row < rows:
            lds[g * 8:(g + 1) * 8] = mat[row, c * 8:(c + 1) * 8]
    return lds


def z_rows_base(lane):
    r, k = lane & 31, zkey(lane & 31)
    return r * ZP + (((lane >> 5) ^ (k & 1)) << 4) + ((k >> 1) << 5)


def z_rows_addr(off, s):
    return off ^ (s << 5)


def z_cols_base(lane):
    r, i = 4 * (lane >> 5) + ((lane & 15) >> 2), lane & 3
    return r * ZP + (((((lane >> 4) & 1) * 2 + (i >> 1)) ^ zkey(r)) << 4) + ((i & 1) << 3)


def z_cols_addr(off, ks, dt):
    a = (off + ks * 16 * ZP) ^ (dt << 6)
    return a, (a ^ 16) + 8 * ZP


def groups(addrs, nbytes):
    g = set()
    for a in addrs:
        for b in range(a, a + nbytes, 16):
            g.add((b // 16) % 16)
    return g


def check(rows=160, verbose=True):
    rng = np.random.default_rng(0)
    mat = rng.integers(0, 65535, size=(rows, 64), dtype=np.uint16)
    lds = build_image(mat)
    ntiles = rows // 32
    worst_rows = worst_cols = 1
    for t in range(ntiles):
        for s in range(4):
            addrs = []
            for lane in range(64):
                a = z_rows_addr(z_rows_base(lane) + t * 32 * ZP, s)
                addrs.append(a)
                got = lds[a // 2:a // 2 + 8]
                c = (2 * s + (lane >> 5)) * 8
                assert (got == mat[t * 32 + (lane & 31), c:c + 8]).all(), ("rows", t, s, lane)
            for grp in B128_GROUPS:      # the four non-contiguous 16-lane groups one LDS cycle serves
                pa = [addrs[l] for l in grp]
                n = len(groups(pa, 16))
                worst_rows = max(worst_rows, 16 // n if n else 99)
                assert n == 16, ("rows conflict", t, s, grp, n)
        for ks in range(2):
            for dt in range(2):
                lo_addrs, hi_addrs = [], []
                for lane in range(64):
                    lo, hi = z_cols_addr(z_cols_base(lane) + t * 32 * ZP, ks, dt)
                    h = lane >> 5
                    rbase = t * 32 + ks * 16 + 4 * h + ((lane & 15) >> 2)
                    col = dt * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4
                    assert (lds[lo // 2:lo // 2 + 4] == mat[rbase, col:col + 4]).all(), ("cols lo", t, ks, dt, lane)
                    assert (lds[hi // 2:hi // 2 + 4] == mat[rbase + 8, col:col + 4]).all(), ("cols hi", t, ks, dt, lane)
                    lo_addrs.append(lo)
                    hi_addrs.append(hi)
                for addrs in (lo_addrs, hi_addrs):
                    for p in range(2):      # passes of 32 lanes, 8 bytes each = 256 B = every bank once
                        pa = addrs[32 * p:32 * p + 32]
                        banks = set()
                        for a in pa:
                            banks.add((a // 4) % 64)
                            banks.add((a // 4 + 1) % 64)
                        assert len(banks) == 64, ("cols conflict", t, ks, dt, p, len(banks))
    if verbose:
        print(f"Z image, {rows} rows: every fragment address returns its logical element; ds_read_b128 lane groups and "
              f"ds_read_b64_tr_b16 passes (32 lanes) are bank-conflict-free")
    return True


def check_rot192(rows=224, verbose=True):
    """The rotated 192-byte-pitch image of csrc/attention_duo.inc (head_dim 96): chunk c of row r at position (c + ((r >> 2) & 3)) mod 12.
    Row fragments by the closed-form per-lane offsets of the kernels (only d-steps 4 and 5 can wrap), column fragments by the
    per-d-tile lo / hi offsets (rotation h / h + 2): logical element + bank checks as in check()."""
    P = 192
    rot = lambda r: (r >> 2) & 3
    rng = np.random.default_rng(1)
    mat = rng.integers(0, 65535, size=(rows, 96), dtype=np.uint16)
    lds = np.zeros(rows * P // 2, dtype=np.uint16)
    for g in range(rows * 12):
        row, pos = g // 12, g % 12
        cc = pos - rot(row)
        cc = cc + 12 if cc < 0 else cc
        lds[g * 8:(g + 1) * 8] = mat[row, cc * 8:(cc + 1) * 8]
    for t in range(rows // 32):
        for st in range(6):
            addrs = []
            for lane in range(64):
                r, h = lane & 31, lane >> 5
                c0 = h + rot(r)
                base = r * P + c0 * 16
                a = base + st * 32 if st < 4 else (base + 128 - (192 if c0 == 4 else 0) if st == 4 else base + 160 - (192 if c0 >= 2 else 0))
                a += t * 32 * P
                c = (2 * st + h) * 8
                assert (lds[a // 2:a // 2 + 8] == mat[t * 32 + r, c:c + 8]).all(), ("rot rows", t, st, lane)
                addrs.append(a)
            for grp in B128_GROUPS:
                assert len(groups([addrs[l] for l in grp], 16)) == 16, ("rot rows conflict", t, st)
        for ks in range(2):
            for d in range(3):
                for half in (0, 1):
                    addrs = []
                    for lane in range(64):
                        ch, cm = lane >> 5, (lane & 15) >> 2
                        cl, csub = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1), ((lane & 3) & 1) * 8
                        pos = (d * 4 + cl + ch + 2 * half) % 12
                        a = t * 32 * P + ks * 16 * P + (4 * ch + cm + 8 * half) * P + pos * 16 + csub
                        r = t * 32 + ks * 16 + 4 * ch + cm + 8 * half
                        col = d * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4
                        assert (lds[a // 2:a // 2 + 4] == mat[r, col:col + 4]).all(), ("rot cols", t, ks, d, half, lane)
                        addrs.append(a)
                    for p_ in range(2):
                        banks = set()
                        for a in addrs[32 * p_:32 * p_ + 32]:
                            banks.add((a // 4) % 64)
                            banks.add((a // 4 + 1) % 64)
                        assert len(banks) == 64, ("rot cols conflict", t, ks, d, half, p_)
    if verbose:
        print(f"rotated 192-byte image, {rows} rows: row and column fragments return their logical elements, bank-conflict-free")
    return True


if __name__ == "__main__":
    check_rot192(224)
    check(160)
    check(224)
