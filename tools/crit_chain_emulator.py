#!/usr/bin/env python3
"""CPU emulator of the critical workgroup's register row-strip chain (csrc/potrf.hip, round 6: factor64_strips).  tools only.

Models the four waves of the workgroup lane by lane (numpy vectors over the 64 lanes), the v_mfma_f64_16x16x4 operand / result lane
maps, and the LDS images the waves exchange, in a valid execution order (a flag that is waited for must have been set: asserted).
It exists to check the INDEX ALGEBRA of the device code (which register of which lane holds which matrix element, which LDS word is
read by whom) against numpy on the host, where a wrong map costs seconds instead of a GPU round trip.  The device code mirrors the
functions below one to one.

  python3 tools/crit_chain_emulator.py            # random SPD 64 x 64 tiles, ragged tiles (nr < 64): |L L^T - D|, |X L - I|, |W - X^T X|
"""
import sys

import numpy as np

LANES = np.arange(64)
I_ = LANES & 15          # row inside the wave's 16-row strip
G_ = LANES >> 4          # k slot / column inside a 4-column block
XB = 17                  # row stride of a 16 x 16 block of X in LDS


def mfma(a, b, c):
    """v_mfma_f64_16x16x4_f64: D[m][n] = C[m][n] + sum_k A[m][k] B[k][n]; lane (m + 16 k) holds A[m][k], lane (n + 16 k) holds B[k][n];
    register q of lane (n + 16 g) holds D[g + 4 q][n]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[I_, G_] = a
    B[G_, I_] = b
    D = A @ B
    out = c.copy()
    for q in range(4):
        out[:, q] += D[G_ + 4 * q, I_]
    return out


class Lds:
    def __init__(self):
        self.xs = np.full(10 * 16 * XB, np.nan)          # packed lower blocks of X, block (r, c) at 16 XB (r (r + 1) / 2 + c)
        self.cpub = np.full(16 * 64, np.nan)             # [kb][jb][16][4]: columns j0 .. j0+3 of the diagonal block, published per 4-column block
        self.lpub = np.full(6 * 256, np.nan)             # L_{r,j}, r > j, as [t][lane] (the register image of its owner)
        self.own = np.full((4, 64), np.nan)              # wave-private exchange buffers
        self.cflag = [0] * 4
        self.lflag = {}
        self.xflag = {}


def xblk(r, c):
    return 16 * XB * (r * (r + 1) // 2 + c)


def lblk(r, j):
    return 256 * (r * (r - 1) // 2 + j)


def pivot_factor(P):
    """the 4 x 4 pivot block (P[r][c], lower entries used): reciprocal square roots and the block's L entries"""
    r0 = 1 / np.sqrt(P[0][0])
    l10, l20, l30 = P[1][0] * r0, P[2][0] * r0, P[3][0] * r0
    d1 = P[1][1] - l10 * l10
    r1 = 1 / np.sqrt(d1)
    l21, l31 = (P[2][1] - l20 * l10) * r1, (P[3][1] - l30 * l10) * r1
    d2 = P[2][2] - l20 * l20 - l21 * l21
    r2 = 1 / np.sqrt(d2)
    l32 = (P[3][2] - l30 * l20 - l31 * l21) * r2
    d3 = P[3][3] - l30 * l30 - l31 * l31 - l32 * l32
    r3 = 1 / np.sqrt(d3)
    return (r0, r1, r2, r3), (l10, l20, l30, l21, l31, l32)


def substitute(p, rs, ls):
    """one row's four block entries p[0..3] against the pivot block -> M[row][0..3] (all 64 lanes at once)"""
    r0, r1, r2, r3 = rs
    l10, l20, l30, l21, l31, l32 = ls
    m0 = p[0] * r0
    m1 = (p[1] - m0 * l10) * r1
    m2 = (p[2] - m0 * l20 - m1 * l21) * r2
    m3 = (p[3] - m0 * l30 - m1 * l31 - m2 * l32) * r3
    return [m0, m1, m2, m3]


def by_g(vals):
    return np.choose(G_, vals)


def chain_diag(lds, kb, a):
    """wave kb: L_kk (in place in a: a[:, t] = D[i][4 t + g] -> L[i][4 t + g], zero above the diagonal); publishes the 16 x 4 column
    block of every 4-column step for the followers; returns rr[:, t] = 1 / L_cc of column c = 4 t + g"""
    rr = np.zeros((64, 4))
    for jb in range(4):
        j0 = 4 * jb
        base = (kb * 4 + jb) * 64
        lds.cpub[base + I_ * 4 + G_] = a[:, jb]
        lds.cflag[kb] = jb + 1
        P = [[lds.cpub[base + (j0 + r) * 4 + c] for c in range(4)] for r in range(4)]
        p = [lds.cpub[base + I_ * 4 + c] for c in range(4)]
        rs, ls = pivot_factor(P)
        m = substitute(p, rs, ls)
        mraw = by_g(m)
        rg = by_g([np.full(64, r) for r in rs])
        mg = np.where(I_ >= j0 + G_, mraw, 0.0)
        if jb < 3:
            a[:] = mfma(-mg, mg, a)
        a[:, jb] = mg
        rr[:, jb] = rg
    return rr


def follow(lds, kb, w, a):
    """wave w > kb: its rows of block column kb, L_{w,kb} (in place in a), one 4-column block behind the diagonal wave"""
    for jb in range(4):
        j0 = 4 * jb
        lds.own[w][I_ * 4 + G_] = a[:, jb]
        assert lds.cflag[kb] > jb, "follower ahead of the diagonal wave"
        base = (kb * 4 + jb) * 64
        P = [[lds.cpub[base + (j0 + r) * 4 + c] for c in range(4)] for r in range(4)]
        qd = [lds.cpub[base + I_ * 4 + c] for c in range(4)]            # the DIAGONAL block's row i
        p = [lds.own[w][I_ * 4 + c] for c in range(4)]                  # this wave's row i
        rs, ls = pivot_factor(P)
        mw = by_g(substitute(p, rs, ls))
        nd = by_g(substitute(qd, rs, ls))
        ng = np.where(I_ >= j0 + G_, nd, 0.0)
        if jb < 3:
            a[:] = mfma(-ng, mw, a)                                     # (D_w^T -= M_diag M_w^T in the accumulator layout = D_w in the a layout)
        a[:, jb] = mw


def invert16(lds, w, a, rr):
    """X_ww = L_ww^-1 from the register image of L_ww: returns x[:, t] = X[4 t + g][i] (the MFMA result layout)"""
    yt = np.zeros((64, 4))
    for t in range(4):
        yt[:, t] = np.where(4 * t + G_ == I_, 1.0, 0.0)
    for jb in range(4):
        j0 = 4 * jb
        lds.own[w][G_ * 16 + I_] = yt[:, jb]
        y = [lds.own[w][k * 16 + I_] for k in range(4)]
        rl = lambda r, c: a[(j0 + r) + 16 * c, jb]                      # v_readlane of a[jb] at lane (i = j0 + r, g = c): L[j0 + r][j0 + c]
        l10, l20, l30, l21, l31, l32 = rl(1, 0), rl(2, 0), rl(3, 0), rl(2, 1), rl(3, 1), rl(3, 2)
        r = [rr[16 * k, jb] for k in range(4)]                          # v_readlane of rr[jb] at lane (0, k)
        w10, w20, w30, w21, w31, w32 = l10 * r[0], l20 * r[0], l30 * r[0], l21 * r[1], l31 * r[1], l32 * r[2]
        z1 = y[1] - w10 * y[0]
        z2 = y[2] - w20 * y[0] - w21 * z1
        z3 = y[3] - w30 * y[0] - w31 * z1 - w32 * z2
        zg = by_g([y[0], z1, z2, z3])
        sg = np.where(I_ > j0 + G_, a[:, jb] * rr[:, jb], 0.0)
        yt = mfma(-sg, zg, yt)
    x = np.zeros((64, 4))
    for t in range(4):
        x[:, t] = np.where(I_ > 4 * t + G_, 0.0, yt[:, t] * rr[:, t])
    return x


def write_x(lds, r, c, x, Xg):
    base = xblk(r, c)
    for q in range(4):
        lds.xs[base + (G_ + 4 * q) * XB + I_] = x[:, q]
        Xg[16 * r + G_ + 4 * q, 16 * c + I_] = x[:, q]
    lds.xflag[(r, c)] = 1


def run_tile(D, nr=64):
    """the whole factor64_strips on one 64 x 64 tile D (lower triangle read); returns L, X, W as the device code stores them"""
    lds = Lds()
    # ---- the a layout after the update phase: wave w, block cb <= w: a[w][cb][:, t] = D[16 w + i][16 cb + 4 t + g], identity padding
    a = [[None] * 4 for _ in range(4)]
    for w in range(4):
        for cb in range(w + 1):
            v = np.zeros((64, 4))
            for t in range(4):
                row, col = 16 * w + I_, 16 * cb + 4 * t + G_
                inside = (row < nr) & (col <= row)
                v[:, t] = np.where(inside, D[np.minimum(row, 63), np.minimum(col, 63)], np.where(row == col, 1.0, 0.0))
            a[w][cb] = v
    Lg = np.zeros((64, 64)); Xg = np.full((64, 64), np.nan); Wg = np.full((64, 64), np.nan)
    rr = [None] * 4
    xdiag = [None] * 4
    # execution order: a valid interleaving of the four wave programs (the asserts check that every wait would have been satisfied)
    for kb in range(4):
        rr[kb] = chain_diag(lds, kb, a[kb][kb])
        for w in range(kb + 1, 4):
            follow(lds, kb, w, a[w][kb])
            # publish L_{w,kb}: the register image as it stands
            for t in range(4):
                lds.lpub[lblk(w, kb) + t * 64 + LANES] = a[w][kb][:, t]
            lds.lflag[(w, kb)] = 1
        for w in range(kb + 1, 4):
            # trailing blocks cb = kb + 1 .. w:  a[cb] -= L_{w,kb} L_{cb,kb}^T in the a layout:  A operand = L_{cb,kb} (published image,
            # or the own registers for cb == w), B operand = the own L_{w,kb}
            for cb in range(kb + 1, w + 1):
                for t in range(4):
                    if cb == w:
                        Aop = a[w][kb][:, t]
                    else:
                        assert lds.lflag[(cb, kb)]
                        Aop = lds.lpub[lblk(cb, kb) + t * 64 + LANES]
                    a[w][cb] = mfma(Aop, -a[w][kb][:, t], a[w][cb])
    # every wave: its strip of L (final) -> global, masked
    for w in range(4):
        for cb in range(w + 1):
            for t in range(4):
                row, col = 16 * w + I_, 16 * cb + 4 * t + G_
                ok = (row < nr) & (col <= row)
                Lg[row[ok], col[ok]] = a[w][cb][ok, t]
    # inverses of the diagonal blocks, the off-diagonal blocks of X row by row (wave c computes column c of every later row)
    for w in range(4):
        xdiag[w] = invert16(lds, w, a[w][w], rr[w])
        write_x(lds, w, w, xdiag[w], Xg)
        for c in range(w + 1, 4):                                         # zero blocks above the diagonal
            for q in range(4):
                Xg[16 * w + G_ + 4 * q, 16 * c + I_] = 0.0
    for r in range(1, 4):
        for c in range(r):                                                # wave c
            T = np.zeros((64, 4))
            for j in range(c, r):
                assert lds.lflag[(r, j)] and lds.xflag[(j, c)]
                for t in range(4):
                    Aop = lds.lpub[lblk(r, j) + t * 64 + LANES]           # lane (m = i, k = g): L_{r,j}[i][4 t + g]
                    Bop = lds.xs[xblk(j, c) + (4 * t + G_) * XB + I_]     # lane (n = i, k = g): X_{j,c}[4 t + g][i]
                    T = mfma(Aop, Bop, T)
            assert lds.xflag[(r, r)]
            X = np.zeros((64, 4))
            for t in range(4):
                Aop = lds.xs[xblk(r, r) + I_ * XB + 4 * t + G_]           # lane (m = i, k = g): X_rr[i][4 t + g]
                X = mfma(-Aop, T[:, t], X)                                # B operand = register t of T (its result layout)
            write_x(lds, r, c, X, Xg)
    # W = X^T X, block (mb, nb), mb >= nb:  sum over rows r >= mb
    owner = {(0, 0): 0, (1, 0): 0, (3, 0): 0, (1, 1): 1, (2, 0): 1, (3, 1): 1, (2, 1): 2, (2, 2): 2, (3, 2): 3, (3, 3): 3}
    for (mb, nb), wv in owner.items():
        acc = np.zeros((64, 4))
        for r in range(mb, 4):
            assert r >= wv
            for t in range(4):
                Aop = lds.xs[xblk(r, mb) + (4 * t + G_) * XB + I_]
                Bop = lds.xs[xblk(r, nb) + (4 * t + G_) * XB + I_]
                acc = mfma(Aop, Bop, acc)
        for q in range(4):
            m, n = 16 * mb + G_ + 4 * q, 16 * nb + I_
            Wg[m, n] = acc[:, q]
            Wg[n, m] = acc[:, q]
    return Lg, Xg, Wg


def main():
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial, nr in enumerate([64, 64, 64, 40, 17, 1, 63, 48]):
        Q = rng.standard_normal((64, 64))
        D = Q @ Q.T / 64 + np.eye(64) * (1e-3 if trial % 2 else 0.5)
        L, X, W = run_tile(np.tril(D), nr)
        Dv = np.eye(64); Dv[:nr, :nr] = D[:nr, :nr]
        Dv = np.tril(Dv)
        Lref = np.linalg.cholesky(Dv + np.tril(Dv, -1).T)
        Lfull = np.eye(64); Lfull[:nr, :nr] = 0
        Lfull = np.tril(L) + np.diag(np.where(np.arange(64) >= nr, 1.0, 0.0))
        e1 = np.abs(Lfull - Lref).max() / np.abs(Lref).max()
        e2 = np.abs(X @ Lref - np.eye(64)).max()
        e3 = np.abs(W - X.T @ X).max() / np.abs(W).max()
        e4 = np.abs(np.triu(X, 1)).max()
        print("nr = %2d: |L - chol| %.1e   |X L - I| %.1e   |W - X^T X| %.1e   upper(X) %.1e   nan %d" % (nr, e1, e2, e3, e4, int(np.isnan(X).sum() + np.isnan(W).sum())))
        worst = max(worst, e1, e3, e4, e2 / max(1.0, np.abs(X).max()))
    print("worst", worst)
    return 0 if worst < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
