#!/usr/bin/env python
"""CPU model of the plane-march PCSOR schedule (petsc_amd/csrc/hipx_sorbox.hip): the index arithmetic, the LDS ring depths and the wait
conditions of the kernel, executed by a randomised scheduler (any runnable agent may take the next step), every ring slot tagged with the
row it holds so a read of a slot that was overwritten or not written yet is an assertion, the result compared with MatSOR_SeqAIJ's loop
(aij.c:1930-1958) entry for entry.  Development tool (no GPU needed): python scripts/sor_box_model.py [nx ny nz] [--rev] [--seed N].

Schedule (logical coordinates; the backward sweep is the forward one on the mirrored grid with the entry order reversed):
  block J, plane k  : lines j = 64 J - k + s, lane s = 0..63 (the block boundaries move one line per plane: every cross-block dependency
                      points to block J - 1, never to J + 1)
  workgroup (J, c)  : planes k = P c + w, wave w = 0..P-1; compute step t: wave w lane s is at row i = t - 2 - 2 s - 4 w
                      (i = -2: idle, i = -1: loads row 0 of its neighbours, 0 <= i < nx: relaxes row i)
  neighbours        : lower plane k - 1 = ring slot w (slot 0 = south plane), line indices s, s+1, s+2;
                      previous line of the same plane = slot w + 1, line index s + 1; index 0, 1 = west lines (from block J - 1 through global x)
  south plane       : (round 6) "wave -1" of the chunk: its virtual step v holds row v + 2 - 2 s of line index s + 2; a POLLER stages pair n = virtual
                      steps 2 n - 2, 2 n - 1 = rows 2 n - 2 s, + 1 (what the chunk below flushed pair by pair: its flush n + 7) when wave 0 is past
                      step 2 n - 17, and wave 0 waits for it as wave w waits for wave w - 1; line indices 0, 1 still come with helper 0's groups
  flushers          : planes 0 .. P - 2 in groups of G steps, the top plane pair by pair
"""
import random
import sys

import numpy as np

P, G = 4, 8
RX, RB, RS, RW = 16, 32, 32, 32  # ring rows: own x lines, right-hand side, south plane, west lines


def reference(nx, ny, nz, coef, dinv, b, rev):
    """MatSOR_SeqAIJ zero-guess sweep on the box stencil (forward: lower entries in CSR order; rev: upper entries, rows descending)."""
    m = nx * ny * nz
    x = np.zeros(m)
    t = np.zeros(m)
    order = sorted(coef)  # (dk, dj, di) ascending = CSR order of the lower part
    rows = range(m - 1, -1, -1) if rev else range(m)
    for r in rows:
        i, j, k = r % nx, (r // nx) % ny, r // (nx * ny)
        s = b[r]
        ents = [(-dk, -dj, -di) for (dk, dj, di) in reversed(order)] if rev else order  # upper part ascending = lower part negated, reversed
        for e, (dk, dj, di) in enumerate(ents):
            ii, jj, kk = i + di, j + dj, k + dk
            if 0 <= ii < nx and 0 <= jj < ny and 0 <= kk < nz:
                c = coef[order[len(order) - 1 - e]] if rev else coef[order[e]]  # symmetric operator in this model: a(r, r + off) = a(r, r - off)
                s = s - c * x[ii + nx * (jj + ny * kk)]
        t[r] = s
        x[r] = s * dinv
    return x, t


class Model:
    def __init__(self, nx, ny, nz, coef, dinv, b, rev, rng):
        self.nx, self.ny, self.nz, self.dinv, self.rev, self.rng = nx, ny, nz, dinv, rev, rng
        self.m = nx * ny * nz
        order = sorted(coef)
        self.ents = [(dk, dj, di, coef[(dk, dj, di)]) for (dk, dj, di) in (reversed(order) if rev else order)]  # logical entries in subtraction order
        self.b = b
        self.gx = [None] * self.m  # global x (None = sentinel), indexed by LOGICAL row
        self.gt = [None] * self.m
        self.nb = (ny + nz - 2) // 64 + 1
        self.nch = (nz + P - 1) // P
        self.T = nx + 2 + 2 * 63 + 4 * (P - 1)
        self.z0 = -0.0 if all(c < 0 for *_, c in self.ents) else 0.0
        assert all(c < 0 for *_, c in self.ents) or all(c > 0 for *_, c in self.ents)

    def lrow(self, i, j, k):  # logical row -> index of b / x in LOGICAL order (the kernel maps to m - 1 - r for the backward sweep)
        return i + self.nx * (j + self.ny * k)

    def run(self):
        # workgroups in ticket order: every dependency of (J, c) is on (J - 1, *) or (J, c - 1), all earlier in this order
        for c in range(self.nch):
            for J in range(self.nb):
                self.run_wg(J, c)

    def run_wg(self, J, c):
        nx, ny, nz = self.nx, self.ny, self.nz
        k0 = P * c
        X = [[[None] * RX for _ in range(66)] for _ in range(P + 1)]  # slots 1..P: own planes (indices 2..65 used: 0, 1 live in W)
        S = [[None] * RS for _ in range(66)]                          # slot 0: south plane
        W = [[[None] * RW for _ in range(2)] for _ in range(P + 1)]   # west lines of slots 1..P
        B = [[[None] * RB for _ in range(64)] for _ in range(P)]
        TR = [[[None] * RX for _ in range(64)] for _ in range(P)]
        cprog = [0] * (P + 2)  # compute steps completed per wave (index w; P, P+1: "no consumer")
        cprog[P] = cprog[P + 1] = 10 ** 9
        hprog = [0] * P        # steps whose inputs are staged (helper w)
        flush = [0] * P        # steps whose outputs are flushed (helper w)
        hgroup = [0] * P
        pbase = [0]            # south poller: pairs staged (virtual steps -2 .. 2 pbase - 3)
        npairs = self.T // 2 + 1 + (self.T % 2)
        ngroups = (self.T + G - 1) // G
        nx2 = nx
        regs = {}

        def line_of(slot, idx):  # logical line (j) and plane of ring slot / line index
            k = k0 + slot - 1
            return 64 * J - k + idx - 2, k

        def gvalue(i, j, k):  # a line of another workgroup through global x (z0 outside the grid)
            if not (0 <= j < ny and 0 <= k < nz and 0 <= i < nx):
                return self.z0
            v = self.gx[self.lrow(i, j, k)]
            assert v is not None, ("halo row not published yet: ticket order broken", J, c, i, j, k)
            return v

        def helper_ready(w):
            g = hgroup[w]
            if g >= ngroups:
                return False
            # ring-overwrite guards: rhs ring RB rows, west ring RW rows (read by wave w and w + 1), south ring RS rows (wave 0)
            if cprog[w] < G * g - (RB - G):
                return False
            if min(cprog[w], cprog[w + 1]) < G * g - (RW - 16):
                return False
            if w == 0 and cprog[0] < G * g - (RS - 16):
                return False
            return True

        def helper_step(w):
            g = hgroup[w]
            k = k0 + w
            for s in range(64):  # (a) right-hand side: rows [8 g - 2 - 2 s - 4 w, + 8)
                j = 64 * J - k + s
                r0 = G * g - 2 - 2 * s - 4 * w
                for r in range(r0, r0 + G):
                    if 0 <= r < nx and 0 <= j < ny and k < nz:
                        B[w][s][r % RB] = (r, self.b[self.lrow(r, j, k)])
            for q in range(2):  # (b) west lines of plane k: rows [8 g - 4 w, + 8)
                j, _ = line_of(w + 1, q)
                r0 = G * g - 4 * w
                for r in range(r0, r0 + G):
                    if 0 <= r < nx:
                        W[w + 1][q][r % RW] = (r, gvalue(r, j, k))
            if w == 0:  # (c) south plane k0 - 1, the two lines in front of block J's: line index q rows [8 g, + 8)
                for q in range(2):
                    j = 64 * J - (k0 - 1) + q - 2
                    r0 = G * g - 2 * max(q - 2, 0)
                    for r in range(r0, r0 + G):
                        if 0 <= r < nx:
                            S[q][r % RS] = (r, gvalue(r, j, k0 - 1))
            hgroup[w] += 1
            hprog[w] = G * (g + 1)

        def fgran(w):  # the top plane leaves pair by pair
            return 2 if w == P - 1 else G

        def flush_ready(w):  # outputs of a group of steps leave once the wave has finished it
            return flush[w] < self.T and cprog[w] >= min(flush[w] + fgran(w), self.T)

        def poller_ready():
            n = pbase[0]
            return n < npairs and cprog[0] >= 2 * n - 17

        def poller_step():
            n = pbase[0]
            for s in range(64):
                j = 64 * J - (k0 - 1) + s
                for r in (2 * n - 2 * s, 2 * n - 2 * s + 1):
                    if 0 <= r < nx:
                        if k0 > 0:
                            rr, _ = S[s + 2][r % RS] if S[s + 2][r % RS] is not None else (None, None)
                            assert rr is None or rr <= r - RS or rr == r, ("south ring overwritten before plane 0 read it", s, r, rr)
                        S[s + 2][r % RS] = (r, gvalue(r, j, k0 - 1))
            pbase[0] = n + 1

        def flush_step(w):
            g = flush[w] // fgran(w)
            k = k0 + w
            for s in range(64):
                j = 64 * J - k + s
                r0 = fgran(w) * g - 2 - 2 * s - 4 * w
                for r in range(r0, r0 + fgran(w)):
                    if 0 <= r < nx and 0 <= j < ny and k < nz:
                        rr, v = X[w + 1][s + 2][r % RX]
                        assert rr == r, ("x ring overwritten before the flush", w, s, r, rr)
                        self.gx[self.lrow(r, j, k)] = v
                        rr, v = TR[w][s][r % RX]
                        assert rr == r
                        self.gt[self.lrow(r, j, k)] = v
            flush[w] = min(flush[w] + fgran(w), self.T)

        def read(slot, idx, r):
            if not 0 <= r < nx:
                return self.z0
            if slot == 0:
                rr, v = S[idx][r % RS]
            elif idx < 2:
                rr, v = W[slot][idx][r % RW]
            else:
                rr, v = X[slot][idx][r % RX]
            assert rr == r, ("ring slot does not hold the row asked for", slot, idx, r, rr)
            return v

        def compute_ready(w):
            t = cprog[w]
            if t >= self.T:
                return False
            if w > 0 and cprog[w - 1] < min(t - 2, self.T):  # lower plane: row i + 1 of line j + 1 was relaxed by wave w - 1 in ITS step t - 3
                return False
            if w == 0 and 2 * pbase[0] < min(t - 2, self.T) + 2:  # the south plane's virtual step t - 3 is staged
                return False
            if hprog[w] <= t:  # inputs of this step staged
                return False
            if cprog[w + 1] < t - 8:  # x ring (RX = 16 rows): wave w + 1 reads row r up to 7 steps after it was written
                return False
            if flush[w] <= t - RX:  # output rings: the rows this step overwrites have left
                return False
            return True

        def compute_step(w):
            t = cprog[w]
            k = k0 + w
            new = {}
            for s in range(64):  # all lanes read first (one instruction each), then compute, then write: lockstep wave
                i = t - 2 - 2 * s - 4 * w
                if i < -1 or i >= nx:
                    continue
                # neighbours' row i + 1: lower plane line indices s, s + 1, s + 2 of slot w; previous line = slot w + 1 index s + 1
                new[s] = [read(w, s, i + 1), read(w, s + 1, i + 1), read(w, s + 2, i + 1), read(w + 1, s + 1, i + 1)]
            for s in range(64):
                i = t - 2 - 2 * s - 4 * w
                if i < -1 or i >= nx:
                    continue
                R = regs.setdefault((w, s), {"L": [[self.z0] * 2 for _ in range(3)], "M": [self.z0] * 2, "xp": self.z0})
                j = 64 * J - k + s
                valid = 0 <= j < ny and k < nz
                if i >= 0:
                    def val(dk, dj, di):
                        if dk == -1:
                            col = R["L"][dj + 1] + [new[s][dj + 1]]
                            return col[di + 1]
                        if dj == -1:
                            return (R["M"] + [new[s][3]])[di + 1]
                        assert (dk, dj, di) == (0, 0, -1)
                        return R["xp"]
                    if valid:
                        rr, sm = B[w][s][i % RB]
                        assert rr == i, ("rhs ring", w, s, i, rr)
                        for (dk, dj, di, cf) in self.ents:
                            sm = sm - cf * val(dk, dj, di)
                        xv = sm * self.dinv
                    else:
                        sm, xv = 0.0, self.z0  # a lane without a line publishes the zero element for its neighbours
                    X[w + 1][s + 2][i % RX] = (i, xv)
                    TR[w][s][i % RX] = (i, sm)
                    R["xp"] = xv
                for l in range(3):
                    R["L"][l] = [R["L"][l][1], new[s][l]]
                R["M"] = [R["M"][1], new[s][3]]
                if i == nx - 1:
                    regs.pop((w, s))
            cprog[w] = t + 1

        agents = [("h", w) for w in range(P)] + [("c", w) for w in range(P)] + [("f", w) for w in range(P)] + [("p", 0)]
        while True:
            runnable = [a for a in agents if {"h": helper_ready, "c": compute_ready, "f": flush_ready, "p": lambda _w: poller_ready()}[a[0]](a[1])]
            if not runnable:
                break
            kind, w = self.rng.choice(runnable)
            {"h": helper_step, "c": compute_step, "f": flush_step, "p": lambda _w: poller_step()}[kind](w)
        assert all(cprog[w] == self.T for w in range(P)) and all(flush[w] == self.T for w in range(P)), ("deadlock", J, c, cprog[:P], hprog, flush)


def main():
    a = [v for v in sys.argv[1:] if not v.startswith("--")]
    nx, ny, nz = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (10, 70, 6)
    rev = "--rev" in sys.argv
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    coef = {}
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                if (dk, dj, di) < (0, 0, 0):
                    coef[(dk, dj, di)] = -float(nprng.integers(1, 9)) / 16.0 - 0.01 * float(nprng.random())
    if "--seven" in sys.argv:
        coef = {k: v for k, v in coef.items() if sum(abs(q) for q in k) == 1}
    dinv = 1.0 / 3.7
    b = nprng.standard_normal(nx * ny * nz)
    xr, tr = reference(nx, ny, nz, coef, dinv, b, rev)
    # the model works in LOGICAL coordinates: the backward sweep is the forward schedule on the mirrored vectors
    M = Model(nx, ny, nz, coef, dinv, b[::-1].copy() if rev else b, rev, rng)
    M.run()
    gx = np.array([v if v is not None else np.nan for v in M.gx])
    gt = np.array([v if v is not None else np.nan for v in M.gt])
    if rev:
        gx, gt = gx[::-1], gt[::-1]
    ok = np.array_equal(gx, xr) and np.array_equal(gt, tr)
    print("box %d x %d x %d  %s  %s entries  blocks %d chunks %d steps %d: %s" % (nx, ny, nz, "backward" if rev else "forward", len(coef), M.nb, M.nch, M.T, "bit-identical" if ok else "DIFFERENT"))
    if not ok:
        bad = np.nonzero(~((gx == xr) & (gt == tr)))[0]
        print("  first differing rows:", bad[:10], "of", len(bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
