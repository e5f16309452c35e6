"""Executable numpy model of the four-wave blocked solve (voxel-slam_amd/csrc/vxba_solve4.hpp): the LDS layout, the block-column
ownership, the panel bookkeeping (which panel has been taken out of which block column), the three-slot g ring, the right-hand-side
chain that runs one step behind the factorisation, and the back substitution -- every index expression is the kernel's.  Between two
barriers the waves' pieces of work are run in several different orders: a result that depends on the order is a race in the kernel.
Checked against numpy.linalg.solve of the damped, gauge-fixed system (Lidar_BA_Optimizer::damping_iter, voxel_map.hpp:397-403) and
against the reference's null-pivot rule (a frame without observations: dxi = 0, the rest of the window moves)."""
import itertools

import numpy as np
import pytest

WAVES, ROW = 4, 6
BLK = 64 * ROW + 6
GSLOT = 64 * ROW


class Cfg:
    def __init__(self, W):
        self.W, self.N, self.M, self.B = W, 6 * W, 6 * W - 6, W - 1
        self.TC = 0
        self.G = self.TC + max(self.B, 1) * BLK
        self.BV = self.G + max(self.B, 1) * GSLOT
        self.ZV = self.BV + 64
        self.XS = self.ZV + 64
        self.DOUBLES = self.XS + 64


def rcp(d):
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        q = 1.0 / d
    return np.where(np.abs(d) > 1e-300, q, 0.0)


def ldl6(D):
    """D: (21, lanes) lower triangle, entry (q, p) at q (q + 1) / 2 + p -> unit lower L in place, inv (6, lanes)."""
    inv = np.zeros((6,) + D.shape[1:])
    ix = lambda q, p: q * (q + 1) // 2 + p
    for p in range(6):
        inv[p] = rcp(D[ix(p, p)])
        col = {q: D[ix(q, p)].copy() for q in range(p + 1, 6)}
        for q in range(p + 1, 6):
            l = col[q] * inv[p]
            for r in range(p + 1, q + 1):
                D[ix(q, r)] = D[ix(q, r)] - l * col[r]
            D[ix(q, p)] = l
    return inv


def solve4_model(H, J, u, W, order_seed=0, return_stats=False):
    """H: gauge-fixed (6W)^2 (rows/cols 0..5 identity), J: gradient.  Returns dxi (6W)."""
    C = Cfg(W)
    n, M, B = C.N, C.M, C.B
    lanes = np.arange(64)
    row_ok = lanes < M
    gi_row = np.where(row_ok, 6 + lanes, 0)
    lds = np.full(C.DOUBLES, np.nan)
    rng = np.random.default_rng(order_seed)

    def load_row(off):            # off: per-lane or scalar base (doubles) -> (6, lanes)
        off = np.broadcast_to(np.asarray(off), (64,))
        return np.stack([lds[off + k] for k in range(6)])

    def store_row(off, a):
        off = np.broadcast_to(np.asarray(off), (64,))
        for k in range(6):
            lds[off + k] = a[k]

    # ---- load (no barrier behind it: a wave stores its first block column at once, the rest -- and the right-hand side -- just before the
    # first step's barrier; wave 0's factorisation of block 0 runs in between)
    def load_block(w, q):
        b = w + WAVES * q
        if b < B:
            a = np.zeros((6, 64))
            for cc in range(6):
                col = 6 + 6 * b + cc
                h = H[gi_row, col]          # Hwork[col * n + row] column-major == H[row, col]
                a[cc] = np.where(row_ok, np.where(col == gi_row, h + u * h, h), 0.0)
            store_row(C.TC + b * BLK + lanes * ROW, a)

    def load_rest():
        for w in range(WAVES):
            for q in range(1, (B + WAVES - 1) // WAVES):
                load_block(w, q)
        lds[C.BV + lanes] = np.where(row_ok, -J[gi_row], 0.0)

    for w in range(WAVES):
        load_block(w, 0)

    ap = [[-1, -1, -1] for _ in range(WAVES)]
    keep = {}
    stats = {"applies": {}, "on_chain": []}

    def apply(t, b, T):
        g = load_row(C.G + t * GSLOT + lanes * ROW)
        lr = C.TC + t * BLK + (6 * b) * ROW
        for c in range(6):
            l = load_row(lr + c * ROW)
            for p in range(6):
                T[c] = T[c] - g[p] * l[p]

    def chain(s):
        w = s & (WAVES - 1)
        sq = s >> 2
        trow = C.TC + s * BLK + lanes * ROW
        a = load_row(trow)
        stats["on_chain"].append(s - 1 - ap[w][sq])
        for t in range(ap[w][sq] + 1, s):
            apply(t, s, a)
        store_row(trow, a)
        dr = C.TC + s * BLK + (6 * s) * ROW
        Ld = np.zeros((21, 64))
        k = 0
        for i in range(6):
            for j in range(i + 1):
                Ld[k] = lds[dr + 6 * i + j]
                k += 1
        inv = ldl6(Ld)
        g = np.zeros((6, 64))
        l = np.zeros((6, 64))
        with np.errstate(invalid="ignore", over="ignore"):
            for p in range(6):
                acc = a[p].copy()
                for q in range(p):
                    acc = acc - g[q] * Ld[p * (p + 1) // 2 + q]
                g[p] = acc
                l[p] = acc * inv[p]
        store_row(trow, l)
        store_row(C.G + s * GSLOT + lanes * ROW, g)
        ap[w][sq] = s
        keep[s] = (Ld, inv, g)

    def rhs(s):
        Ld, inv, g = keep.pop(s)
        bb = load_row(C.BV + 6 * s)
        br = lds[C.BV + lanes].copy()
        z = np.zeros((6, 64))
        with np.errstate(invalid="ignore", over="ignore"):
            for p in range(6):
                acc = bb[p].copy()
                for q in range(p):
                    acc = acc - bb[q] * Ld[p * (p + 1) // 2 + q]
                bb[p] = acc
                z[p] = acc * inv[p]
            for p in range(6):
                br = br - g[p] * z[p]
        d = lanes - 6 * s
        zsel = z[0].copy()
        for p in range(1, 6):
            zsel = np.where(d == p, z[p], zsel)
        inblk = (d >= 0) & (d < 6)
        lds[C.ZV + lanes[inblk]] = zsel[inblk]
        lds[C.BV + lanes[~inblk]] = br[~inblk]

    def catch_up(w, s):
        budget = 2
        for q in range((B + WAVES - 1) // WAVES):
            b = w + WAVES * q
            must = b == s + 2
            if b > s and b < B and ap[w][q] < s and (must or budget > 0):
                trow = C.TC + b * BLK + lanes * ROW
                T = load_row(trow)
                with np.errstate(invalid="ignore", over="ignore"):
                    while ap[w][q] < s and (must or budget > 0):
                        apply(ap[w][q] + 1, b, T)
                        ap[w][q] += 1
                        budget -= 1
                        stats["applies"][s] = stats["applies"].get(s, {})
                        stats["applies"][s][w] = stats["applies"][s].get(w, 0) + 1
                store_row(trow, T)

    # interval -1: chain(0); then for every s: barrier s, {rhs(s) + catch-up(s) of the non-next-owners} run beside chain(s + 1)
    if B > 0:
        chain(0)
    load_rest()
    for s in range(B):
        owner = s & (WAVES - 1)
        nxt = (s + 1) & (WAVES - 1) if s + 1 < B else -1
        per_wave = {}
        for w in range(WAVES):
            tasks = []
            if w == owner:
                tasks.append(lambda s=s: rhs(s))
            if w != nxt:
                tasks.append(lambda w=w, s=s: catch_up(w, s))
            else:
                tasks.append(lambda s=s: chain(s + 1))
            per_wave[w] = tasks
        for w in rng.permutation(WAVES):     # whole waves in a random order: the extreme interleavings
            for t in per_wave[w]:
                t()

    # ---- back substitution (wave 0)
    x = np.where(row_ok, lds[C.ZV + lanes], 0.0)
    if B > 0:
        jl = np.where(row_ok, lanes, 0)
        lcol = C.TC + (jl // 6) * BLK + jl % 6
        for rb in range(B - 1, -1, -1):
            for k in range(6):
                r = 6 * rb + 5 - k
                Lr = lds[lcol + r * ROW]
                xr = x[r]
                upd = lanes < r
                x = np.where(upd, x - np.where(upd, Lr, 0.0) * xr, x)
    dxi = np.zeros(n)
    dxi[6:] = x[:M]
    if return_stats:
        return dxi, stats
    return dxi


def make_system(W, seed, cond=1e4, null_frame=None):
    rng = np.random.default_rng(seed)
    n = 6 * W
    A = rng.standard_normal((3 * n, n)) * np.logspace(0, np.log10(cond) / 2, n)[None, :]
    H = A.T @ A
    J = rng.standard_normal(n)
    if null_frame is not None:
        sl = slice(6 * null_frame, 6 * null_frame + 6)
        H[sl, :] = 0.0
        H[:, sl] = 0.0
        J[sl] = 0.0
    # gauge fix of the reference (voxel_map.hpp:397-400): frame 0's rows and columns become the identity, its gradient 0
    H[:6, :] = 0.0
    H[:, :6] = 0.0
    H[:6, :6] = np.eye(6)
    J[:6] = 0.0
    return H, J


@pytest.mark.parametrize("W", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_model_solves_the_damped_system(W):
    u = 0.01
    H, J = make_system(W, seed=100 + W)
    ref = np.zeros(6 * W)
    if W > 1:
        Hd = H + u * np.diag(np.diag(H))
        ref[6:] = np.linalg.solve(Hd[6:, 6:], -J[6:])
    for order_seed in range(6):
        got = solve4_model(H, J, u, W, order_seed)
        assert np.all(np.isfinite(got))
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (W, order_seed)


def test_model_result_does_not_depend_on_the_wave_order():
    H, J = make_system(10, seed=7)
    base = solve4_model(H, J, 0.3, 10, 0)
    for order_seed in range(1, 12):
        assert np.array_equal(base, solve4_model(H, J, 0.3, 10, order_seed))


@pytest.mark.parametrize("null_frame", [1, 4, 9])
def test_model_null_pivot_rule(null_frame):
    """A frame that observes nothing: all-zero block row / column; Eigen's LDLT::solve returns 0 for it (voxel_map.hpp:403)."""
    W, u = 10, 0.01
    H, J = make_system(W, seed=3, null_frame=null_frame)
    got = solve4_model(H, J, u, W, 1)
    keep = np.ones(6 * W, bool)
    keep[:6] = False
    keep[6 * null_frame:6 * null_frame + 6] = False
    Hd = H + u * np.diag(np.diag(H))
    ref = np.zeros(6 * W)
    ref[keep] = np.linalg.solve(Hd[np.ix_(keep, keep)], -J[keep])
    assert np.all(np.isfinite(got))
    assert np.all(got[6 * null_frame:6 * null_frame + 6] == 0.0)
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_model_schedule_keeps_the_backlog_off_the_chain():
    """The catch-up policy (two (block, panel) updates per wave and interval, the block that is urgent next first): the owner of a block
    always finds exactly one panel left to take out of it, and no wave does more than two updates between two barriers."""
    H, J = make_system(10, seed=11)
    _, st = solve4_model(H, J, 0.01, 10, 3, return_stats=True)
    assert st["on_chain"][0] == 0 and all(k == 1 for k in st["on_chain"][1:])
    assert max(n for per in st["applies"].values() for n in per.values()) <= 2
