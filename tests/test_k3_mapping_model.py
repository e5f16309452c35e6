"""Executable model of the Hessian sweep's work mapping (voxel-slam_amd/csrc/vxba_k3.hpp + k3_finalize_kernel), in numpy.

The kernel cannot run without a GPU, but everything in it that is index arithmetic can be checked here: the run of batches a
workgroup takes, the step / ragged-last-step logic, the row-major tile and its row stride, which wave multiplies which pairs of
4-column groups (k3_make_pairs), the lane maps of v_mfma_f64_4x4x4_4b_f64 with its four blocks used as four K-slices of one
output block (probed on the MI355X: scripts/ubench/k3_blockk_probe.hip -- operand lane 16 k + 4 t + i, result lane 16 i + 4 t + j),
the fold over the four blocks, the 512-byte store pattern of the epilogue, the workgroup partial's layout and k3_finalize's
assembly.  The model mirrors the kernel's formulas one for one (same names) and is compared with the plain definition
H = -B^T B + blockdiag(D)."""
import numpy as np
import pytest

DACC = 28
WAVES = 8


def k3_groups(W):
    return (6 * W + 3) // 4


def k3_row_stride(W):
    rs = 4 * k3_groups(W)
    while rs % 8 != 4:
        rs += 1
    return rs


def k3_nv(W):
    n = min(64 // W, 12) & ~1
    while n > 2 and 2 * 24 * n * k3_row_stride(W) * 8 > 140 * 1024:
        n -= 2
    return n


def k3_make_pairs(NG):
    """vxba_k3.hpp k3_make_pairs: per wave the (I, J) pairs, their operand slots and the column group of every slot."""
    L = []

    def intra(s0, s1):
        for i in range(s0, s1):
            for j in range(i, s1):
                L.append((i, j))

    def rows(r0, r1, c0, c1):
        for i in range(r0, r1):
            for j in range(c0, c1):
                L.append((i, j))
    if NG == 15:
        intra(0, 5); intra(5, 10); intra(10, 15)
        rows(0, 3, 5, 10)
        rows(3, 5, 5, 10); rows(3, 4, 10, 15)
        rows(0, 3, 10, 15)
        rows(4, 5, 10, 15); rows(5, 7, 10, 15)
        rows(7, 10, 10, 15)
    else:
        for a in range(0, NG, 5):
            intra(a, min(a + 5, NG))
        for a in range(0, NG, 5):
            for b in range(a + 5, NG, 5):
                rows(a, a + 5, b, min(b + 5, NG))
    NP = NG * (NG + 1) // 2
    assert len(L) == NP and len(set(L)) == NP and all(i <= j for i, j in L)
    PPW = (NP + 7) // 8
    tab = []
    for w in range(8):
        pairs = L[w * PPW:min((w + 1) * PPW, NP)]
        slot_group, sa, sb = [], [], []
        for (i, j) in pairs:
            for gidx in (i, j):
                if gidx not in slot_group:
                    slot_group.append(gidx)
            sa.append(slot_group.index(i)); sb.append(slot_group.index(j))
        tab.append(dict(pairs=pairs, slot_group=slot_group, sa=sa, sb=sb))
    return tab, PPW


class Cfg:
    def __init__(s, W):
        s.W = W
        s.NG = k3_groups(W)
        s.NP = s.NG * (s.NG + 1) // 2
        s.NCOLS = 4 * s.NG
        s.RS = k3_row_stride(W)
        s.NV = k3_nv(W)
        s.R = 3 * s.NV
        s.ROWS = WAVES * s.R
        s.KC = s.ROWS // 16
        s.BUF = s.ROWS * s.RS
        s.tab, s.PPW = k3_make_pairs(s.NG)
        s.PPWP = (s.PPW + 3) & ~3
        s.NTILE = WAVES * s.PPWP * 16
        s.PLEN = s.NTILE + W * DACC
        assert s.NV % 2 == 0 and s.NV * W <= 64 and s.ROWS % 16 == 0 and s.RS % 8 == 4 and s.RS >= s.NCOLS
        assert s.NG <= 15 and s.PPW <= 16 and max(len(t["slot_group"]) for t in s.tab) <= 16
        # two tile buffers + poses + LM inputs + parameter staging + dump within the CU's LDS
        assert (2 * s.BUF + 24 * W + 8 + WAVES * s.NV * 18 + 32) * 8 <= 160 * 1024
        assert 512 * (DACC + 1) * 8 <= 160 * 1024

    def npair(s, w):
        return max(0, min(s.PPW, s.NP - w * s.PPW))


def test_geometry_of_every_window_size():
    for W in range(1, 11):
        C = Cfg(W)
        assert sum(len(t["pairs"]) for t in C.tab) == C.NP
        for w in range(8):
            assert len(C.tab[w]["pairs"]) == C.npair(w)
    C = Cfg(10)
    assert (C.NG, C.NP, C.RS, C.NV, C.ROWS, C.KC, C.PPW, C.PLEN) == (15, 120, 60, 6, 144, 9, 15, 2048 + 280)
    # the hand-made order at W = 10: operand reads per slab and wave
    assert [len(t["slot_group"]) for t in C.tab] == [5, 5, 5, 8, 12, 8, 8, 8]


def test_operand_reads_are_bank_conflict_free():
    """ds_read_b64 is served 32 lanes at a time over 64 banks of 4 bytes: the 32 lanes of a half-wave must touch 64 distinct banks."""
    for W in range(1, 11):
        C = Cfg(W)
        for half in range(2):
            banks = []
            for lane in range(32 * half, 32 * half + 32):
                byte = ((lane >> 2) * C.RS + (lane & 3)) * 8
                banks += [(byte // 4) % 64, (byte // 4 + 1) % 64]
            assert len(set(banks)) == 64, (W, half)


def sym6_index(a, b):
    return b if a == 0 else (2 + b if a == 1 else 5)


def mfma_f64_4x4x4_4b(a, b):
    """v_mfma_f64_4x4x4_4b_f64 as probed: operand lane l = 16 k + 4 t + i holds A_t[i][k] (B: B_t[k][j] with j for i);
    result lane 16 i + 4 t + j holds D_t[i][j] = sum_k A_t[i][k] B_t[k][j]."""
    d = np.zeros(64)
    for t in range(4):
        A = np.zeros((4, 4)); B = np.zeros((4, 4))
        for k in range(4):
            for i in range(4):
                A[i, k] = a[16 * k + 4 * t + i]
                B[k, i] = b[16 * k + 4 * t + i]
        D = A @ B
        for i in range(4):
            for j in range(4):
                d[16 * i + 4 * t + j] = D[i, j]
    return d


def run_model(W, V, head, end, G, rng):
    C = Cfg(W)
    n = 6 * W
    rows = rng.normal(size=(V, W, 3, 6))
    lin = rng.normal(size=(V, W, DACC))
    obs = rng.random(size=(V, W)) < 0.8
    rows[~obs] = 0.0
    lin[~obs] = 0.0

    # ---- reference: H = -B^T B + blockdiag(D), JacT, residual over voxels [head, end)
    H = np.zeros((n, n)); J = np.zeros(n); res = 0.0
    for a in range(head, end):
        B = np.zeros((3, n))
        for i in range(W):
            B[:, 6 * i:6 * i + 6] = rows[a, i]
        H -= B.T @ B
        for i in range(W):
            d = lin[a, i]
            J[6 * i:6 * i + 6] += d[0:6]
            D = np.zeros((6, 6))
            for x in range(3):
                for y in range(x, 3):
                    D[x, y] = D[y, x] = d[6 + sym6_index(x, y)]
                    D[3 + x, 3 + y] = D[3 + y, 3 + x] = d[21 + sym6_index(x, y)]
                for y in range(3):
                    D[x, 3 + y] = d[12 + 3 * x + y]
                    D[3 + y, x] = d[12 + 3 * x + y]
            H[6 * i:6 * i + 6, 6 * i:6 * i + 6] += D
            if i == 0:
                res += d[27]

    # ---- the kernel, workgroup by workgroup
    partial = np.full((G, C.PLEN), np.nan)          # slots nobody writes must never be read by k3_finalize
    b0, b1 = head // C.NV, (end - 1) // C.NV
    nb_all = b1 - b0 + 1
    q, rem = nb_all // G, nb_all % G
    covered = []
    for g in range(G):
        cnt = q + (1 if g < rem else 0)
        bs = b0 + g * q + min(g, rem)
        covered += list(range(bs, bs + cnt))
        lds = np.full((2, C.BUF), np.nan)            # whatever is read must have been written
        for b in range(2):                           # prologue: the padding columns 6W .. 4 NG
            for rr in range(C.ROWS):
                lds[b, rr * C.RS + 6 * W: rr * C.RS + C.NCOLS] = 0.0
        acc = np.zeros((WAVES, C.PPW, 64))           # [wave][pair j][lane]
        dacc = np.zeros((WAVES, 64, DACC))

        def phase_m(buf, nch):
            for wave in range(WAVES):
                T = C.tab[wave]
                for qq in range(nch):
                    x = np.zeros((max(1, len(T["slot_group"])), 64))
                    for lane in range(64):
                        opnd = (lane >> 2) * C.RS + (lane & 3)
                        for s_, grp in enumerate(T["slot_group"]):
                            x[s_, lane] = buf[opnd + qq * 16 * C.RS + 4 * grp]
                    for j in range(len(T["pairs"])):
                        acc[wave, j] += mfma_f64_4x4x4_4b(x[T["sa"][j]], x[T["sb"][j]])

        def phase_a(buf, wave, b):
            for lane in range(C.NV * W):
                vl, fi = lane // W, lane % W
                a = b * C.NV + vl
                ok = head <= a < end
                rws = rows[a, fi] if ok and a < V else np.zeros((3, 6))
                if ok:
                    dacc[wave, lane] += lin[a, fi]
                for r in range(3):
                    o = (wave * C.R + 3 * vl + r) * C.RS + 6 * fi
                    buf[o:o + 6] = rws[r]

        nfull, nrag = cnt // WAVES, cnt % WAVES
        for s in range(nfull + 1):
            if s >= 1:
                phase_m(lds[(s - 1) & 1], C.KC)
            if s == nfull:
                break
            for wave in range(WAVES):
                phase_a(lds[s & 1], wave, bs + s * WAVES + wave)
        if nrag > 0:
            buf = lds[nfull & 1]
            nch = (nrag * C.R + 15) >> 4
            for wave in range(WAVES):
                if wave < nrag:
                    phase_a(buf, wave, bs + nfull * WAVES + wave)
                elif wave == nrag:
                    buf[nrag * C.R * C.RS: nch * 16 * C.RS] = 0.0
            phase_m(buf, nch)
        assert not np.isnan(acc).any()
        # epilogue
        pout = partial[g]
        for el in range(W * DACC):
            i, k = el // DACC, el % DACC
            pout[C.NTILE + el] = sum(dacc[w_, v * W + i, k] for w_ in range(WAVES) for v in range(C.NV))
        for wave in range(WAVES):
            v = acc[wave].copy()
            lanes = np.arange(64)
            v = v + v[:, lanes ^ 4]
            v = v + v[:, lanes ^ 8]
            np_w = C.npair(wave)
            for m in range(C.PPWP // 4):
                for lane in range(64):
                    t4 = (lane >> 2) & 3
                    if 4 * m + t4 < np_w:
                        off = (wave * C.PPWP + 4 * m + t4) * 16 + 4 * (lane >> 4) + (lane & 3)
                        assert np.isnan(pout[off])            # every slot is stored once
                        pout[off] = v[4 * m + t4, lane]
    assert covered == list(range(b0, b1 + 1))

    # ---- k3_finalize (fin_map)
    Hk = np.zeros((n, n)); Jk = np.zeros(n); resk = 0.0
    seen = np.zeros((n, n), dtype=int)
    tot = partial.sum(axis=0)
    for e in range(C.PLEN):
        if e < C.NTILE:
            wv, pl, i, j = e // (16 * C.PPWP), (e >> 4) % C.PPWP, (e >> 2) & 3, e & 3
            if pl >= C.npair(wv):
                continue
            I, Jg = C.tab[wv]["pairs"][pl]
            r, c = 4 * I + i, 4 * Jg + j
            if r >= n or c >= n or r > c:
                continue
            t1 = 0.0
            if r // 6 == c // 6:
                fr, a, b = r // 6, r % 6, c % 6
                if b < 3: d = 6 + sym6_index(a, b)
                elif a < 3: d = 12 + 3 * a + (b - 3)
                else: d = 21 + sym6_index(a - 3, b - 3)
                t1 = tot[C.NTILE + fr * DACC + d]
            assert not np.isnan(tot[e])
            Hk[r, c] = Hk[c, r] = t1 - tot[e]
            seen[r, c] += 1
        else:
            qq = e - C.NTILE; i, d = qq // DACC, qq % DACC
            if d < 6: Jk[6 * i + d] = tot[e]
            elif d == 27 and i == 0: resk = tot[e]
    assert (seen[np.triu_indices(n)] == 1).all()      # every upper-triangular entry comes from exactly one partial element
    return (H, J, res), (Hk, Jk, resk)


@pytest.mark.parametrize("W,V,head,end,G", [
    (10, 333, 0, 333, 4),      # several full steps + a ragged one per workgroup
    (10, 100, 7, 95, 3),       # sub-range not aligned to batches
    (10, 5, 0, 5, 4),          # fewer batches than workgroups
    (9, 200, 0, 200, 2),       # odd W: two zero padding columns
    (8, 150, 3, 150, 2),
    (7, 120, 0, 120, 2),
    (5, 250, 0, 250, 2),
    (6, 170, 0, 170, 3),
    (4, 300, 0, 300, 2),
    (3, 400, 0, 400, 2),
    (2, 300, 0, 300, 2),
    (1, 300, 10, 290, 2),
])
def test_k3_work_mapping_reproduces_the_definition(W, V, head, end, G):
    rng = np.random.default_rng(100 * W + V)
    (H, J, res), (Hk, Jk, resk) = run_model(W, V, head, end, G, rng)
    sc = np.abs(H).max()
    assert np.allclose(Hk, H, rtol=0, atol=1e-11 * sc)
    assert np.allclose(Jk, J, rtol=1e-12, atol=1e-12)
    assert np.isclose(resk, res, rtol=1e-12)
