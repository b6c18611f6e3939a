"""CPU: the batch programs libhipx compiles (csrc/hipx_pipe.hip: kBatchProgs) are the update blocks of the reference's pipelined loops, statement by statement.
Each block is written here as the reference writes it (kind, y, x by NAME, in source order); the canonical slot numbering (for each operation y, then x, numbered
by first appearance -- what the drop-in's lazy queue produces, plugin/vechipx.c VecHIPXLazyTryBatch) must be a program hipxVecBatchProgramKnown() accepts, and
where /root/reference exists every statement must appear in the reference's source in that order."""
import ctypes as C
import os
import re

import pytest

REF = "/root/reference/src/ksp/ksp/impls"
BLOCKS = {
    # KSPSolve_PIPECG, i > 0 (pipecg.c:140-150)
    "pipecg": ("cg/pipecg/pipecg.c", [("AYPX", "Z", "N"), ("AYPX", "Q", "M"), ("AYPX", "P", "U"), ("AYPX", "S", "W"), ("AXPY", "X", "P"), ("AXPY", "U", "Q"), ("AXPY", "W", "Z"), ("AXPY", "R", "S")]),
    # KSPSolve_PIPECG, i == 0: the four VecCopy in front are not recorded
    "pipecg_first": ("cg/pipecg/pipecg.c", [("AXPY", "X", "P"), ("AXPY", "U", "Q"), ("AXPY", "W", "Z"), ("AXPY", "R", "S")]),
    # KSPSolve_GROPPCG (groppcg.c:98-100 and 135-136)
    "groppcg_a": ("cg/groppcg/groppcg.c", [("AXPY", "x", "p"), ("AXPY", "r", "s"), ("AXPY", "z", "S")]),
    "groppcg_b": ("cg/groppcg/groppcg.c", [("AYPX", "p", "z"), ("AYPX", "s", "Z")]),
    # KSPSolve_PIPECR (pipecr.c:108-117)
    "pipecr": ("cr/pipecr/pipecr.c", [("AYPX", "Z", "N"), ("AYPX", "Q", "M"), ("AYPX", "P", "U"), ("AXPY", "X", "P"), ("AXPY", "U", "Q"), ("AXPY", "W", "Z")]),
}


def canonical(ops):
    slot, kind, ys, xs = {}, [], [], []
    for k, y, x in ops:
        for v in (y, x):
            slot.setdefault(v, len(slot))
        kind.append(1 if k == "AXPY" else 2)
        ys.append(slot[y])
        xs.append(slot[x])
    return kind, ys, xs, len(slot)


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_library_knows_the_reference_update_block(built, name):
    from petsc_amd import _lib
    hx, _ = _lib.load()
    kind, ys, xs, nvec = canonical(BLOCKS[name][1])
    n = len(kind)
    arr = lambda v: (C.c_int * n)(*v)  # noqa: E731
    assert hx.hipxVecBatchProgramKnown(n, arr(kind), arr(ys), arr(xs), nvec) == 1
    # ... and not a permutation of it where the order matters (PIPECG / PIPECR: x += a p must see the NEW p; blocks of independent operations canonicalise to
    # the same program in any order -- and give the same vectors)
    if name in ("pipecg", "pipecr"):
        k2, y2, x2, nv2 = canonical(BLOCKS[name][1][::-1])
        assert hx.hipxVecBatchProgramKnown(n, arr(k2), arr(y2), arr(x2), nv2) == 0


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not here")
@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_is_the_reference_source_in_order(name):
    path, ops = BLOCKS[name]
    txt = open(os.path.join(REF, path)).read()
    pos = txt.index("do {")
    for k, y, x in ops:
        m = re.compile(r"Vec%s\(%s,\s*[-\w]+,\s*%s\)" % (k, y, x)).search(txt, pos)
        assert m, (name, k, y, x)
        pos = m.end()
