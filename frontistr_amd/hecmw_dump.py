"""Reader / writer of the reference's matrix dump files (`!SOLVER, DUMPTYPE=BSR`):
dump_matrix_<call>_<rank>.bsr / .rhs / .sol as written by hecmw_mat_dump_bsr, hecmw_mat_dump_rhs and
hecmw_mat_dump_solution (hecmw1/src/solver/matrix/hecmw_matrix_dump.f90:235-369).  Format: header lines starting
with '%', `nrow ncol nnonzero ndof`, row index (0:nrow), 1-based block column per block (lower, diagonal, upper of
each row), then one value per line (e20.12e3), NDOF*NDOF per block, row-major.  This is the golden-vector channel
of SURVEY §2: systems dumped by a real fistr1 run can be replayed through hecmw_solve here."""
import numpy as np

from . import hecmw


def _numbers(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("%"):
                yield line


def read_bsr(path):
    """-> hecmwST_matrix with D / AL / AU split out of the merged rows (B, X zero)."""
    it = _numbers(path)
    nrow, ncol, nnz, ndof = (int(t) for t in next(it).split())
    if ndof != 3:
        raise NotImplementedError("only 3x3 blocks are on the hot path (NDOF=%d in %s)" % (ndof, path))
    index = np.array([int(next(it)) for _ in range(nrow + 1)], dtype=np.int64)
    item = np.array([int(next(it)) for _ in range(nnz)], dtype=np.int32)
    val = np.array([float(next(it)) for _ in range(nnz * 9)], dtype=np.float64).reshape(nnz, 9)
    rows = np.repeat(np.arange(1, nrow + 1), np.diff(index))
    low, dia, upp = item < rows, item == rows, item > rows
    if dia.sum() != nrow:
        raise ValueError("%s: every row needs exactly one diagonal block" % path)
    indexL = np.zeros(nrow + 1, dtype=np.int32)
    indexU = np.zeros(nrow + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[low], minlength=nrow + 1)[1:], out=indexL[1:])
    np.cumsum(np.bincount(rows[upp], minlength=nrow + 1)[1:], out=indexU[1:])
    return hecmw.hecmwST_matrix.from_arrays(nrow, ncol, indexL, item[low], indexU, item[upp], val[dia].ravel(),
                                            val[low].ravel(), val[upp].ravel())


def read_vector(path):
    return np.array([float(t) for t in _numbers(path)], dtype=np.float64)


def write_bsr(path, m):
    """The same file hecmw_mat_dump_bsr writes (values with 12 digits)."""
    with open(path, "w") as f:
        n = m.NP
        nnz = n + int(m.indexL[n]) + int(m.indexU[n])
        f.write("%%Block-CSR matrix real general\n% nrow ncol nnonzero ndof\n")
        f.write("%d %d %d %d\n%% index(0:nrow)\n0\n" % (n, m.NP, nnz, 3))
        idx = 0
        for i in range(n):
            idx += int(m.indexL[i + 1] - m.indexL[i]) + 1 + int(m.indexU[i + 1] - m.indexU[i])
            f.write("%d\n" % idx)
        f.write("% item(1:nnonzero)\n")
        for i in range(n):
            for j in range(m.indexL[i], m.indexL[i + 1]):
                f.write("%d\n" % m.itemL[j])
            f.write("%d\n" % (i + 1))
            for j in range(m.indexU[i], m.indexU[i + 1]):
                f.write("%d\n" % m.itemU[j])
        f.write("% value(1:nnonzero*ndof*ndof)\n")

        def blk(a, k):
            for v in a[9 * k:9 * k + 9]:
                f.write(_e20(v) + "\n")
        for i in range(n):
            for j in range(m.indexL[i], m.indexL[i + 1]):
                blk(m.AL, j)
            blk(m.D, i)
            for j in range(m.indexU[i], m.indexU[i + 1]):
                blk(m.AU, j)


def _e20(v):
    """Fortran e20.12e3: 0.dddddddddddd E+eee, right-justified in 20 columns."""
    if v == 0.0:
        return "0.000000000000E+000".rjust(20)
    s = "%.11E" % v                                  # d.ddddddddddd E+xx
    mant, exp = s.split("E")
    sign = "-" if mant.startswith("-") else ""
    digits = mant.lstrip("-").replace(".", "")       # 12 digits
    return ("%s0.%sE%+04d" % (sign, digits, int(exp) + 1)).rjust(20)
