"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libhecmw_oracle.so (the CPU
restatement of the reference).  Import only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HALO_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)


class _Matrix(C.Structure):
    _fields_ = [("N", C.c_int32), ("NP", C.c_int32),
                ("indexL", C.c_void_p), ("itemL", C.c_void_p),
                ("indexU", C.c_void_p), ("itemU", C.c_void_p),
                ("D", C.c_void_p), ("AL", C.c_void_p), ("AU", C.c_void_p)]


class _Comm(C.Structure):
    _fields_ = [("halo", HALO_FN), ("allreduce", ALLREDUCE_FN), ("ctx", C.c_void_p)]


def build():
    subprocess.run(["make", "-C", HERE, "libhecmw_oracle.so"], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libhecmw_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_inner_product.restype = C.c_double
        L.orc_precond_setup.restype = C.c_void_p
        L.orc_precond_perm.restype = C.POINTER(C.c_int32)
        L.orc_precond_colorindex.restype = C.POINTER(C.c_int32)
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def cmatrix(m):
    """m: any object with N, NP, indexL, itemL, indexU, itemU, D, AL, AU (refrun.BSR)."""
    s = _Matrix(m.N, m.NP, _p(m.indexL), _p(m.itemL), _p(m.indexU), _p(m.itemU),
                _p(m.D), _p(m.AL), _p(m.AU))
    s._keep = m
    return s


class Comm:
    """Python-side halo/allreduce hooks (used by the gloo multi-process tests)."""

    def __init__(self, halo=None, allreduce=None, nvec=0):
        self.nvec = nvec

        def _h(x, _ctx):
            if halo is not None:
                halo(np.ctypeslib.as_array(x, shape=(nvec,)))

        def _a(v, n, _ctx):
            if allreduce is not None:
                allreduce(np.ctypeslib.as_array(v, shape=(n,)))

        self._h, self._a = HALO_FN(_h), ALLREDUCE_FN(_a)
        self.c = _Comm(self._h, self._a, None)

    def ref(self):
        return C.byref(self.c)


def matvec(m, x, comm=None):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    y = np.zeros(3 * m.NP)
    A = cmatrix(m)
    lib().orc_matvec_33(C.byref(A), comm.ref() if comm else None, _dp(x), _dp(y))
    return y


class Precond:
    def __init__(self, m, precond, sigma_diag=1.0, ncolor_in=10, nthreads=1):
        self.m = m
        self.A = cmatrix(m)
        self.h = C.c_void_p(lib().orc_precond_setup(C.byref(self.A), precond, C.c_double(sigma_diag),
                                                     ncolor_in, nthreads))
        if not self.h:
            raise ValueError("unsupported precond %d" % precond)

    def apply(self, r, iterpremax=1, comm=None):
        r = np.ascontiguousarray(r, dtype=np.float64).copy()
        z = np.zeros(3 * self.m.NP)
        zp = np.zeros(3 * self.m.NP)
        lib().orc_precond_apply(C.byref(self.A), comm.ref() if comm else None, self.h, iterpremax,
                                _dp(r), _dp(z), _dp(zp))
        return z

    @property
    def ncolor(self):
        return lib().orc_precond_ncolor(self.h)

    @property
    def perm(self):
        return np.ctypeslib.as_array(lib().orc_precond_perm(self.h), shape=(self.m.N,)).copy()

    @property
    def colorindex(self):
        return np.ctypeslib.as_array(lib().orc_precond_colorindex(self.h), shape=(self.ncolor + 1,)).copy()

    def __del__(self):
        try:
            if self.h:
                lib().orc_precond_free(self.h)
                self.h = None
        except Exception:
            pass


def solve_iterative(m, I, R, nthreads=1, comm=None):
    """orc_solve_iterative: returns dict(code, iter, resid, history, X, Iarray)."""
    I = np.ascontiguousarray(I, dtype=np.int32).copy()
    R = np.ascontiguousarray(R, dtype=np.float64).copy()
    X = np.ascontiguousarray(m.X, dtype=np.float64).copy()
    B = np.ascontiguousarray(m.B, dtype=np.float64)
    hist = np.zeros(max(int(I[0]), 1))
    it, rs = C.c_int(0), C.c_double(0.0)
    A = cmatrix(m)
    code = lib().orc_solve_iterative(C.byref(A), comm.ref() if comm else None, _dp(B), _dp(X), _ip(I),
                                     _dp(R), nthreads, C.byref(it), C.byref(rs), _dp(hist))
    n = min(it.value, hist.size)
    return dict(code=code, iter=it.value, resid=rs.value, history=hist[:n].copy(), X=X, Iarray=I)


def mat_con(NP, conn):
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    n_elem, nn = conn.shape
    indexL = np.zeros(NP + 1, dtype=np.int32)
    indexU = np.zeros(NP + 1, dtype=np.int32)
    lib().orc_mat_con(NP, n_elem, nn, _ip(conn), _ip(indexL), _ip(indexU), None, None)
    itemL = np.zeros(max(indexL[NP], 1), dtype=np.int32)
    itemU = np.zeros(max(indexU[NP], 1), dtype=np.int32)
    lib().orc_mat_con(NP, n_elem, nn, _ip(conn), _ip(indexL), _ip(indexU), _ip(itemL), _ip(itemU))
    return indexL, itemL[:indexL[NP]].copy(), indexU, itemU[:indexU[NP]].copy()


def stf_c3d8(elemopt, ecoord, E, nu):
    ec = np.ascontiguousarray(ecoord, dtype=np.float64).reshape(8, 3)
    k = np.zeros((24, 24))
    lib().orc_stf_c3d8(elemopt, _dp(ec), C.c_double(E), C.c_double(nu), _dp(k))
    return k


def assemble(elemopt, coord, conn, E, nu, bc=None, load=None):
    """Profile + element loop + Dirichlet BC, returning a refrun.BSR-like object."""
    from .refrun import BSR
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    NP = coord.shape[0]
    indexL, itemL, indexU, itemU = mat_con(NP, conn)
    D = np.zeros(9 * NP)
    AL = np.zeros(9 * max(itemL.size, 1))
    AU = np.zeros(9 * max(itemU.size, 1))
    lib().orc_assemble_c3d8(elemopt, NP, conn.shape[0], _dp(coord), _ip(conn), C.c_double(E),
                            C.c_double(nu), _ip(indexL), _ip(itemL), _ip(indexU), _ip(itemU),
                            _dp(D), _dp(AL), _dp(AU))
    B = np.zeros(3 * NP) if load is None else np.ascontiguousarray(load, dtype=np.float64).copy()
    if bc is not None:
        node, dof, val = bc
        for n_, d_, v_ in zip(node, dof, val):
            lib().orc_mat_ass_bc(NP, _ip(indexL), _ip(itemL), _ip(indexU), _ip(itemU), _dp(D), _dp(AL),
                                 _dp(AU), _dp(B), int(n_), int(d_), C.c_double(v_))
    return BSR(NP, NP, indexL, itemL, indexU, itemU, D, AL[:9 * itemL.size], AU[:9 * itemU.size], B)
