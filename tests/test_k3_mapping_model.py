"""Executable model of the Hessian sweep's work mapping (voxel-slam_amd/csrc/vxba_k3.hpp + k3_finalize_kernel), in numpy.

The kernel cannot run without a GPU, but everything in it that is index arithmetic can be checked here: the run of batches a
workgroup takes, the step / ragged-last-step logic, the pair-row tile layout `at(row, col)`, which wave multiplies which K range of
which tile pairs, the MFMA lane maps, the epilogue's parking order, the workgroup partial's layout and k3_finalize's assembly --
including the block-diagonal terms Drt / Dtt that the matrix cores deliver through the spare columns.  The model mirrors the
kernel's formulas one for one (same names) and is compared with the plain definition  H = -B^T B + blockdiag(D)."""
import numpy as np
import pytest

DACC = 28
WAVES = 8


class Cfg:
    def __init__(s, W):
        s.W = W
        s.NT = (6 * W + 15) // 16
        s.NTP = s.NT * (s.NT + 1) // 2
        s.NCOL = 16 * s.NT
        cap = 12 if s.NT <= 2 else (8 if s.NT == 3 else 6)
        s.NV = min(64 // W, cap)
        s.R = 3 * s.NV
        s.ROWS = WAVES * s.R
        s.KS = s.ROWS // 4
        s.TSPLIT = 2 if s.NTP >= 6 else 1
        s.KSPLIT = WAVES // s.TSPLIT
        s.TPW = s.NTP // s.TSPLIT
        s.KPW = s.KS // s.KSPLIT
        s.SPARE = (s.NCOL - 6 * W) >= 3
        s.BUF = s.ROWS * s.NCOL
        assert s.ROWS % 4 == 0 and s.NTP % s.TSPLIT == 0 and s.KS % s.KSPLIT == 0
        assert 2 * s.BUF * 8 + 12 * W * 8 <= 160 * 1024

    def at(s, row, col):
        return (row >> 1) * 2 * s.NCOL + (col >> 4) * 32 + (row & 1) * 16 + (col & 15)

    def _upper(s, t):
        I = 0
        while t >= s.NT - I:
            t -= s.NT - I
            I += 1
        return I, I + t

    # operand slots: tile j of a wave multiplies slot pa(j) (rows) with slot pb(j) (columns); the column tile a slot reads depends on the set
    @property
    def NSLOT(s):
        return 4 if s.NT == 4 else (6 if s.NT == 3 else s.NT)

    def pa(s, j):
        return (j if j < 3 else j - 2) if s.NT == 4 else (2 * j if s.NT == 3 else s._upper(j)[0])

    def pb(s, j):
        return (j + 1 if j < 3 else j - 2) if s.NT == 4 else (2 * j + 1 if s.NT == 3 else s._upper(j)[1])

    def slot_tile(s, st, k):
        if s.NT == 4:
            return k if st == 0 else (2, 0, 3, 1)[k]
        if s.NT == 3:
            return (0, 0, 0, 1, 1, 1)[k] if st == 0 else (2, 2, 0, 2, 1, 2)[k]
        return k

    def rowtile(s, t):
        return s.slot_tile(t // s.TPW, s.pa(t % s.TPW))

    def coltile(s, t):
        return s.slot_tile(t // s.TPW, s.pb(t % s.TPW))

    def elem_offset(s, r, c):
        I, J = r >> 4, c >> 4
        for t in range(s.NTP):
            if s.rowtile(t) == I and s.coltile(t) == J:
                row, col = r - 16 * I, c - 16 * J
            elif s.rowtile(t) == J and s.coltile(t) == I:
                row, col = c - 16 * J, r - 16 * I
            else:
                continue
            return t * 256 + (row >> 2) * 64 + ((row & 3) << 4) + col
        return -1


def sym6_index(a, b):
    return b if a == 0 else (2 + b if a == 1 else 5)


def run_model(W, V, head, end, G, rng):
    C = Cfg(W)
    n = 6 * W
    # per (voxel, frame): rows of B (3 x 6), linear accumulators; per voxel: the three spare values
    rows = rng.normal(size=(V, W, 3, 6))
    lin = rng.normal(size=(V, W, DACC))
    if C.SPARE:
        lin[:, :, 12:27] = 0.0     # Drt / Dtt do not exist as accumulators
    spare = rng.normal(size=(V, 3))
    # the z row is what Drt / Dtt are made of: give it the structure the kernel's mathematics has, z = sz [w ; n u], spare = kappa u
    u = rng.normal(size=(V, 3)); kappa = rng.normal(size=V); sz = rng.normal(size=(V, W)); nn = rng.normal(size=(V, W)); w = rng.normal(size=(V, W, 3))
    rows[:, :, 2, 0:3] = sz[:, :, None] * w
    rows[:, :, 2, 3:6] = (sz * nn)[:, :, None] * u[:, None, :]
    spare = kappa[:, None] * u
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
            if C.SPARE:   # what the spare columns deliver: Drt = sum z[0:3] spare^T, Dtt = sum z[3:6] spare^T
                zr = rows[a, i, 2]
                D[0:3, 3:6] += np.outer(zr[0:3], spare[a]); D[3:6, 0:3] += np.outer(zr[0:3], spare[a]).T
                D[3:6, 3:6] += np.outer(zr[3:6], spare[a])
            H[6 * i:6 * i + 6, 6 * i:6 * i + 6] += D
            if i == 0:
                res += d[27]

    # ---- the kernel, workgroup by workgroup
    PLEN = C.NTP * 256 + W * DACC
    partial = np.zeros((G, PLEN))
    b0, b1 = head // C.NV, (end - 1) // C.NV
    nb_all = b1 - b0 + 1
    q, rem = nb_all // G, nb_all % G
    covered = []
    for g in range(G):
        cnt = q + (1 if g < rem else 0)
        bs = b0 + g * q + min(g, rem)
        covered += list(range(bs, bs + cnt))
        lds = np.full((2, C.BUF), 0.0)
        acc = np.zeros((WAVES, C.TPW, 4, 64))          # [wave][tile j][register r][lane]
        dacc = np.zeros((WAVES, 64, DACC))

        def phase_m(buf, nb_prev):
            for wave in range(WAVES):
                st, kq = wave % C.TSPLIT, wave // C.TSPLIT
                if nb_prev >= WAVES:
                    k0, nk = kq * C.KPW, C.KPW
                else:
                    ks = (nb_prev * C.R + 3) >> 2
                    k0 = (kq * ks) // C.KSPLIT
                    nk = ((kq + 1) * ks) // C.KSPLIT - k0
                for kk in range(nk):
                    x = np.zeros((C.NSLOT, 64))
                    for lane in range(64):
                        lrow, lcol = lane >> 4, lane & 15
                        for k in range(C.NSLOT):
                            x[k, lane] = buf[C.at(4 * k0 + lrow, lcol) + 32 * C.slot_tile(st, k) + kk * 4 * C.NCOL]
                    for j in range(C.TPW):
                        # v_mfma_f64_16x16x4: A[i][k] in lane 16k+i, B[k][jj] in lane 16k+jj, D[(l/16)+4r][l%16] in register r of lane l
                        A = x[C.pa(j)].reshape(4, 16).T      # [i][k]
                        Bm = x[C.pb(j)].reshape(4, 16)       # [k][jj]
                        D = A @ Bm
                        for lane in range(64):
                            for r in range(4):
                                acc[wave, j, r, lane] += D[(lane >> 4) + 4 * r, lane & 15]

        def phase_a(buf, wave, b):
            for lane in range(C.NV * W):
                vl, fi = lane // W, lane % W
                a = b * C.NV + vl
                ok = head <= a < end
                rws = rows[a, fi] if ok and a < V else np.zeros((3, 6))
                if ok:
                    dacc[wave, lane] += lin[a, fi]
                for r in range(3):
                    for jj in range(3):
                        o = C.at(wave * C.R + 3 * vl + r, 0) + C.at(0, 6 * fi + 2 * jj)
                        buf[o] = rws[r, 2 * jj]; buf[o + 1] = rws[r, 2 * jj + 1]
                if C.SPARE and fi == W - 1:
                    o = C.at(wave * C.R + 3 * vl + 2, 0) + C.at(0, 6 * W)
                    sp = spare[a] if a < V else np.ones(3)    # an out-of-range slot carries some finite voxel's values
                    buf[o], buf[o + 1], buf[o + 2] = sp

        nfull, nrag = cnt // WAVES, cnt % WAVES
        for s in range(nfull + 1):
            if s >= 1:
                phase_m(lds[(s - 1) & 1], WAVES)
            if s == nfull:
                break
            for wave in range(WAVES):
                phase_a(lds[s & 1], wave, bs + s * WAVES + wave)
        if nrag > 0:
            buf = lds[nfull & 1]
            for wave in range(WAVES):
                if wave < nrag:
                    phase_a(buf, wave, bs + nfull * WAVES + wave)
                elif wave == nrag:
                    z0 = C.at(nrag * C.R, 0)
                    buf[z0:z0 + 4 * C.NCOL] = 0.0
            phase_m(buf, nrag)
        # epilogue
        pout = partial[g]
        for el in range(W * DACC):
            i, k = el // DACC, el % DACC
            pout[C.NTP * 256 + el] = sum(dacc[w_, v * W + i, k] for w_ in range(WAVES) for v in range(C.NV))
        for el in range(C.NTP * 256):
            t, x_ = el >> 8, el & 255
            ts, j = t // C.TPW, t % C.TPW
            pout[el] = sum(acc[k * C.TSPLIT + ts, j, x_ >> 6, x_ & 63] for k in range(C.KSPLIT))
    assert covered == list(range(b0, b1 + 1))

    # ---- k3_finalize
    NTILE = C.NTP * 256
    Hk = np.zeros((n, n)); Jk = np.zeros(n); resk = 0.0
    tot = partial.sum(axis=0)
    for e in range(PLEN):
        if e < NTILE:
            t, j, l = e >> 8, (e >> 6) & 3, e & 63
            rt, ct = C.rowtile(t), C.coltile(t)
            r = 16 * rt + (l >> 4) + 4 * j
            c = 16 * ct + (l & 15)
            if rt > ct:
                r, c = c, r
            if r >= n or c >= n or r > c:
                continue
            t1 = 0.0
            if r // 6 == c // 6:
                i, a, b = r // 6, r % 6, c % 6
                if C.SPARE and b >= 3:
                    off1 = C.elem_offset(r, n + (b - 3))
                    assert off1 >= 0
                else:
                    if b < 3: d = 6 + sym6_index(a, b)
                    elif a < 3: d = 12 + 3 * a + (b - 3)
                    else: d = 21 + sym6_index(a - 3, b - 3)
                    off1 = NTILE + i * DACC + d
                t1 = tot[off1]
            Hk[r, c] = Hk[c, r] = t1 - tot[e]
        else:
            qq = e - NTILE; i, d = qq // DACC, qq % DACC
            if d < 6: Jk[6 * i + d] = tot[e]
            elif d == 27 and i == 0: resk = tot[e]
    return (H, J, res), (Hk, Jk, resk)


@pytest.mark.parametrize("W,V,head,end,G", [
    (10, 333, 0, 333, 4),      # several full steps + a ragged one per workgroup
    (10, 100, 7, 95, 3),       # sub-range not aligned to batches
    (10, 5, 0, 5, 4),          # fewer batches than workgroups
    (9, 200, 0, 200, 2),
    (8, 150, 3, 150, 2),       # no spare columns: register accumulators
    (5, 250, 0, 250, 2),       # no spare columns
    (6, 170, 0, 170, 3),
    (3, 400, 0, 400, 2),
    (1, 300, 10, 290, 2),
])
def test_k3_work_mapping_reproduces_the_definition(W, V, head, end, G):
    rng = np.random.default_rng(100 * W + V)
    (H, J, res), (Hk, Jk, resk) = run_model(W, V, head, end, G, rng)
    sc = np.abs(H).max()
    assert np.allclose(Hk, H, rtol=0, atol=1e-11 * sc)
    assert np.allclose(Jk, J, rtol=1e-12, atol=1e-12)
    assert np.isclose(resk, res, rtol=1e-12)
