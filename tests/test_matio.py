"""CPU: the matrix-file readers behind `bench.py --matrix-file` (BASELINE config 4 ships as a SuiteSparse MatrixMarket file; PETSc's
own exchange format is the binary Mat file) against a file the REFERENCE wrote and against the reference's MatLoad."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
from petsc_amd import matio

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")


def test_petsc_binary_written_by_the_reference():
    """tests/golden/ref_mat_7pt_n4.bin = `ref_driver -stencil 7 -n 4 -mat_view binary:<file>` (MatView_SeqAIJ_Binary of the reference
    build in oracle/_ref): the reader returns the operator the oracle assembles, bit for bit."""
    ai, aj, aa = matio.read_matrix(os.path.join(HERE, "golden", "ref_mat_7pt_n4.bin"))
    bi, bj, ba = orc.stencil("7pt", 4)
    assert np.array_equal(ai, bi) and np.array_equal(aj, bj) and np.array_equal(aa, ba)
    assert aj.dtype == np.int32 and aa.dtype == np.float64


def test_petsc_binary_round_trip(tmp_path):
    ai, aj, aa = orc.stencil("27pt", 5)
    aa = aa * (1.0 + np.arange(aa.size) / 7.0)
    p = str(tmp_path / "a.bin")
    matio.write_petsc_binary(p, ai, aj, aa)
    ci, cj, ca = matio.read_matrix(p)
    assert np.array_equal(ai, ci) and np.array_equal(aj, cj) and np.array_equal(aa, ca)
    with open(p, "r+b") as f:  # a wrong class id is refused
        f.write(b"\x00\x00\x00\x01")
    with pytest.raises(ValueError):
        matio.read_petsc_binary(p)


def test_matrix_market_kinds(tmp_path):
    """general with a repeated entry (summed in file order), symmetric (lower triangle mirrored, mmloader.c:91-104), skew-symmetric,
    pattern, integer; comment lines and blank lines; gzip."""
    g = tmp_path / "g.mtx"
    g.write_text("%%MatrixMarket matrix coordinate real general\n% a comment\n\n3 3 5\n1 1 2.0\n3 2 0.5\n1 1 0.25\n2 2 -1e0\n3 3 7\n")
    ai, aj, aa = matio.read_matrix(str(g))
    assert ai.tolist() == [0, 1, 2, 4] and aj.tolist() == [0, 1, 1, 2] and aa.tolist() == [2.25, -1.0, 0.5, 7.0]
    s = tmp_path / "s.mtx"
    s.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2.0\n2 1 -1.0\n3 3 5\n3 2 0.5\n")
    ai, aj, aa = matio.read_matrix(str(s))
    assert ai.tolist() == [0, 2, 4, 6] and aj.tolist() == [0, 1, 0, 2, 1, 2] and aa.tolist() == [2.0, -1.0, -1.0, 0.5, 0.5, 5.0]
    k = tmp_path / "k.mtx"
    k.write_text("%%MatrixMarket matrix coordinate integer skew-symmetric\n2 2 1\n2 1 3\n")
    ai, aj, aa = matio.read_matrix(str(k))
    assert ai.tolist() == [0, 1, 2] and aj.tolist() == [1, 0] and aa.tolist() == [-3.0, 3.0]
    pz = tmp_path / "p.mtx.gz"
    with gzip.open(str(pz), "wt") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 2\n2 1\n")
    ai, aj, aa = matio.read_matrix(str(pz))
    assert aj.tolist() == [1, 0] and aa.tolist() == [1.0, 1.0]
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError):
        matio.read_matrix(str(bad))


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref is not built (needs /root/reference)")
def test_reference_matload_reads_what_we_write(tmp_path):
    """A MatrixMarket file -> matio -> PETSc binary -> the REFERENCE's MatLoad + KSPSolve (`ref_driver -f`): the same residual history,
    digit for digit, as the reference assembling that operator itself."""
    ai, aj, aa = orc.stencil("7pt", 6)
    rows = np.repeat(np.arange(len(ai) - 1), np.diff(ai))
    low = aj <= rows
    mtx = tmp_path / "p7.mtx"
    with open(str(mtx), "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%d %d %d\n" % (len(ai) - 1, len(ai) - 1, int(low.sum())))
        for r, c, v in zip(rows[low], aj[low], aa[low]):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))
    bi, bj, ba = matio.read_matrix(str(mtx))
    assert np.array_equal(bi, ai) and np.array_equal(bj, aj) and np.array_equal(ba, aa)
    binf = str(tmp_path / "p7.bin")
    matio.write_petsc_binary(binf, bi, bj, ba)
    common = ["-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-10", "-history"]
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    a = subprocess.run([REF, "-f", binf] + common, stdout=subprocess.PIPE, text=True, env=env, check=True).stdout
    b = subprocess.run([REF, "-stencil", "7", "-n", "6"] + common, stdout=subprocess.PIPE, text=True, env=env, check=True).stdout
    ha = [l for l in a.splitlines() if l.startswith("hist ")]
    hb = [l for l in b.splitlines() if l.startswith("hist ")]
    assert len(ha) > 5 and ha == hb
