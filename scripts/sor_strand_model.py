#!/usr/bin/env python
"""SIMT model (numpy, CPU) of the strand-scheduled SOR sweep of petsc_amd/csrc/hipx_sor.hip (sor_strand_kernel).

Design tool, not product code: it executes the kernel's algorithm -- one lane per strand (a run of L consecutive rows, an
x-line of the grid), 64 strands per wave, per-lane progress pointers, a tagged LDS window per wave for the values of the
lanes' own strands and of the "far" strands staged from global memory, a modelled global-memory visibility latency -- and
checks that (a) the result is bit-identical to the sequential sweep (oracle) and (b) how many wave iterations the critical
path takes.  Usage: python scripts/sor_strand_model.py [7|27] [n] [glat]
"""
import sys

import numpy as np

sys.path.insert(0, "oracle")
sys.path.insert(0, ".")
import oracle as orc  # noqa: E402

WP = 16      # window positions per LDS row
SD = 4       # staging pipeline depth (iterations between issue and use of a far load)
LA = 6       # staging look-ahead beyond the stager lane's own position
SENT = None


def templates(ai, aj, aa):
    rows = {}
    tid = np.zeros(len(ai) - 1, np.int32)
    tl = []
    for r in range(len(ai) - 1):
        key = tuple((int(aj[k] - r), float(aa[k])) for k in range(ai[r], ai[r + 1]))
        if key not in rows:
            rows[key] = len(tl)
            tl.append(key)
        tid[r] = rows[key]
    return tid, tl


def pick_L(tl, counts):
    t = tl[int(np.argmax(counts))]
    offs = sorted(o for o, _ in t)
    clusters = [[offs[0]]]
    for o in offs[1:]:
        if o == clusters[-1][-1] + 1:
            clusters[-1].append(o)
        else:
            clusters.append([o])
    centers = [(c[0] + c[-1]) // 2 for c in clusters]
    pos = [c for c in centers if c > 0]
    return min(pos) if pos else None


def sweep_model(ai, aj, aa, b, idiag, forward, glat, x_init_t=None, told=None, xold=None, omega=1.0):
    """KIND 0 (forward, zero guess: t = b - L x, x = t idiag) or KIND 1 (backward: x = (1-w) xold + (t - U x) idiag)."""
    m = len(ai) - 1
    tid, tl = templates(ai, aj, aa)
    counts = np.bincount(tid, minlength=len(tl))
    L = pick_L(tl, counts)
    assert L, "no strand length"
    nstr = (m + L - 1) // L
    # dependency entries per template in arithmetic order, logical (ds, dp)
    dep = []
    for t in tl:
        ent = []
        for off, v in t:
            if (off < 0) if forward else (off > 0):
                lo = off if forward else -off
                ds = int(np.floor(lo / L + 0.5))
                dp = lo - ds * L
                ent.append((ds, dp, v))
        dep.append(ent)
    dsall = sorted({e[0] for ent in dep for e in ent} | {0})
    bands = [[dsall[0], dsall[0]]]
    for d in dsall[1:]:
        if d == bands[-1][1] + 1:
            bands[-1][1] = d
        else:
            bands.append([d, d])
    band_of = {}
    rowbase = []
    nr = 0
    for bi, (lo, hi) in enumerate(bands):
        rowbase.append(nr)
        nr += 64 + hi - lo
        for d in range(lo, hi + 1):
            band_of[d] = bi
    npanels = (nstr + 63) // 64
    xnew = np.full(m, np.nan)
    pub_tick = np.full(m, 1 << 60, np.int64)
    tvec = np.zeros(m)

    def actual(q):
        return q if forward else m - 1 - q

    class Wave:
        pass

    waves = []
    for P in range(npanels):
        w = Wave()
        w.S0 = P * 64
        w.p = np.zeros(64, np.int64)
        w.len = np.array([max(0, min(L, m - (w.S0 + l) * L)) if w.S0 + l < nstr else 0 for l in range(64)])
        w.val = np.zeros((nr, WP))
        w.tag = np.full((nr, WP), -1, np.int64)
        # staging duties: (band, which) -> per lane row u, strand, frontier, issue pointer, pipeline
        w.duty = []
        for bi, (lo, hi) in enumerate(bands):
            for which in (0, 1):
                u = np.arange(64) + 64 * which
                strand = w.S0 + u + lo
                active = (u < 64 + hi - lo) & ((strand < w.S0) | (strand > w.S0 + 63)) & (strand >= 0) & (strand < nstr)
                if active.any():
                    d = Wave()
                    d.row = rowbase[bi] + u
                    d.strand = strand
                    d.active = active
                    d.sf = np.zeros(64, np.int64)   # next position to accept
                    d.si = np.zeros(64, np.int64)   # next position to issue
                    d.pipe = [[None] * 64 for _ in range(SD)]  # (pos, value or None)
                    d.slen = np.array([max(0, min(L, m - s * L)) if 0 <= s < nstr else 0 for s in strand])
                    d.cons = [[c for c in range(uu - (hi - lo), uu + 1) if 0 <= c < 64] for uu in u]  # lanes that read this row
                    w.duty.append(d)
        w.iters = 0
        w.done = False
        waves.append(w)

    tick = 0
    stalls = 0
    while not all(w.done for w in waves):
        tick += 1
        assert tick < 200000
        for w in waves:
            if w.done:
                continue
            w.iters += 1
            # 1. staging: consume the oldest in-flight load of every duty, then issue a new one
            for d in w.duty:
                old = d.pipe.pop(0)
                d.pipe.append([None] * 64)
                for l in range(64):
                    if not d.active[l]:
                        continue
                    it = old[l]
                    if it is not None:
                        pos, v = it
                        if pos == d.sf[l]:
                            if v is not None:
                                w.val[d.row[l], pos % WP] = v
                                w.tag[d.row[l], pos % WP] = pos
                                d.sf[l] += 1
                            else:  # not published yet: flush the pipeline, restart from the frontier
                                d.si[l] = d.sf[l]
                                for st in d.pipe:
                                    st[l] = None
                    lead = max([w.p[c] for c in d.cons[l] if w.p[c] < w.len[c]] or [-LA - 1])  # most advanced consumer still running
                    tgt = min(d.slen[l], lead + LA + 1)
                    if d.si[l] < tgt and d.si[l] < d.sf[l] + WP - 4:
                        q = d.strand[l] * L + d.si[l]
                        r = actual(q)
                        v = xnew[r] if pub_tick[r] + glat <= tick else None
                        d.pipe[-1][l] = (int(d.si[l]), v)
                        d.si[l] += 1
            # 2. tight step (all LDS reads first)
            newvals = {}
            for l in range(64):
                if w.p[l] >= w.len[l]:
                    continue
                S = w.S0 + l
                q = S * L + w.p[l]
                r = actual(q)
                ok = True
                vals = []
                for ds, dp, a in dep[tid[r]]:
                    pos = w.p[l] + dp
                    assert 0 <= pos < L, "template entry leaves the strand"
                    u = rowbase[band_of[ds]] + l + ds - bands[band_of[ds]][0]
                    tg = w.tag[u, pos % WP]
                    if tg == pos:
                        vals.append(w.val[u, pos % WP])
                    elif tg > pos:  # overwritten: direct global read
                        rr = actual((S + ds) * L + pos)
                        if pub_tick[rr] + glat <= tick:
                            vals.append(xnew[rr])
                        else:
                            ok = False
                            break
                    else:
                        ok = False
                        break
                if not ok:
                    stalls += 1
                    continue
                if forward:
                    s = b[r]
                    for (ds, dp, a), v in zip(dep[tid[r]], vals):
                        s -= a * v
                    tvec[r] = s
                    out = s * idiag[r]
                else:
                    s = told[r]
                    for (ds, dp, a), v in zip(dep[tid[r]], vals):
                        s -= a * v
                    out = (1 - omega) * xold[r] + s * idiag[r]
                newvals[l] = (r, out)
            # 3. publish
            for l, (r, out) in newvals.items():
                for bi, (lo, hi) in enumerate(bands):  # every window row that shows this lane's strand (small grids: several bands)
                    u = l - lo
                    if 0 <= u < 64 + hi - lo:
                        w.val[rowbase[bi] + u, w.p[l] % WP] = out
                        w.tag[rowbase[bi] + u, w.p[l] % WP] = w.p[l]
                xnew[r] = out
                pub_tick[r] = tick
                w.p[l] += 1
            if (w.p >= w.len).all():
                w.done = True
    return xnew, tvec, tick, max(w.iters for w in waves), stalls, L, bands


def main():
    kind = {"7": "7pt", "27": "27pt"}[sys.argv[1] if len(sys.argv) > 1 else "27"]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    glat = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ai, aj, aa = orc.stencil(kind, n)
    m = len(ai) - 1
    rng = np.random.default_rng(1)
    b = rng.standard_normal(m)
    diag = np.array([aa[k] for r in range(m) for k in range(ai[r], ai[r + 1]) if aj[k] == r])
    idiag = 1.0 / diag
    x1, t1, ticks, iters, stalls, L, bands = sweep_model(ai, aj, aa, b, idiag, True, glat)
    x2, _, ticks2, iters2, stalls2, _, _ = sweep_model(ai, aj, aa, b, idiag, False, glat, told=t1, xold=x1)
    import ctypes as C
    xo = np.zeros(m)
    orc.lib().orc_MatSOR_SeqAIJ(m, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(b), C.c_double(1.0), 12 | 16, C.c_double(0.0), 1, 1, orc.P(xo))
    print("kind %s n %d L %d bands %s glat %d: fwd ticks %d (max wave iters %d, stalls %d), bwd ticks %d; levels ideal ~%d; bit-exact %s" %
          (kind, n, L, bands, glat, ticks, iters, stalls, ticks2, n + 2 * n + 4 * n if kind == "27pt" else 3 * n, np.array_equal(x2, xo)))
    assert np.array_equal(x2, xo)


if __name__ == "__main__":
    main()
