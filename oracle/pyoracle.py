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
                ("D", C.c_void_p), ("AL", C.c_void_p), ("AU", C.c_void_p), ("ndof", C.c_int32)]


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
                _p(m.D), _p(m.AL), _p(m.AU), int(getattr(m, "NDOF", 3)))
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
    y = np.zeros(int(getattr(m, "NDOF", 3)) * m.NP)
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
        z = np.zeros(int(getattr(self.m, "NDOF", 3)) * self.m.NP)
        zp = np.zeros(z.size)
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
    hist = np.zeros(max(int(I[0]), 1) + 2)      # GMRES may log MAXIT+1 lines
    it, rs = C.c_int(0), C.c_double(0.0)
    A = cmatrix(m)
    code = lib().orc_solve_iterative(C.byref(A), comm.ref() if comm else None, _dp(B), _dp(X), _ip(I),
                                     _dp(R), nthreads, C.byref(it), C.byref(rs), _dp(hist))
    n = min(it.value, hist.size) if int(I[1]) == 3 else min(it.value, int(I[0]))   # a DO loop that runs out leaves ITER = MAXIT+1
    return dict(code=code, iter=it.value, resid=rs.value, history=hist[:n].copy(), X=X, Iarray=I)


def solve_sequence(m, I, R, nsolve, nthreads=1, growth=1.1):
    """The recycle policy of the preconditioner over a sequence of solves, as oracle/ref_solve_driver.f90 mode 4 runs it through
    the reference: solve; then (D *= growth, X = X0, Iarray(97) = 1, Iarray(98) = 0, solve) nsolve - 1 times.
    Returns (iteration counts, last X, Iarray)."""
    from .refrun import BSR
    I = np.ascontiguousarray(I, dtype=np.int32).copy()
    D = np.ascontiguousarray(m.D, dtype=np.float64).copy()
    lib().orc_persist_precond(1)
    iters = []
    try:
        for k in range(nsolve):
            if k > 0:
                D = D * growth
                I[96], I[97] = 1, 0
            Ak = BSR(m.N, m.NP, m.indexL, m.itemL, m.indexU, m.itemU, D, m.AL, m.AU, m.B, m.X, NDOF=getattr(m, "NDOF", 3))
            o = solve_iterative(Ak, I, R, nthreads=nthreads)
            I = o["Iarray"]
            iters.append(o["iter"])
    finally:
        lib().orc_persist_precond(0)
    return iters, o["X"], I


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


def assemble(elemopt, coord, conn, E, nu, bc=None, load=None, sections=None):
    """Profile + element loop + Dirichlet BC, returning a refrun.BSR-like object.  sections = (E[], nu[], elem_mat[]
    1-based) assembles several materials (E, nu ignored)."""
    from .refrun import BSR
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    NP = coord.shape[0]
    indexL, itemL, indexU, itemU = mat_con(NP, conn)
    D = np.zeros(9 * NP)
    AL = np.zeros(9 * max(itemL.size, 1))
    AU = np.zeros(9 * max(itemU.size, 1))
    if sections is None:
        lib().orc_assemble_c3d8(elemopt, NP, conn.shape[0], _dp(coord), _ip(conn), C.c_double(E),
                                C.c_double(nu), _ip(indexL), _ip(itemL), _ip(indexU), _ip(itemU),
                                _dp(D), _dp(AL), _dp(AU))
    else:
        Es = np.ascontiguousarray(sections[0], dtype=np.float64)
        nus = np.ascontiguousarray(sections[1], dtype=np.float64)
        em = np.ascontiguousarray(sections[2], dtype=np.int32)
        lib().orc_assemble_c3d8_sections(elemopt, NP, conn.shape[0], _dp(coord), _ip(conn), _dp(Es), _dp(nus), _ip(em),
                                         _ip(indexL), _ip(itemL), _ip(indexU), _ip(itemU), _dp(D), _dp(AL), _dp(AU))
    B = np.zeros(3 * NP) if load is None else np.ascontiguousarray(load, dtype=np.float64).copy()
    if bc is not None:
        node, dof, val = bc
        for n_, d_, v_ in zip(node, dof, val):
            lib().orc_mat_ass_bc(NP, _ip(indexL), _ip(itemL), _ip(indexU), _ip(itemU), _dp(D), _dp(AL),
                                 _dp(AU), _dp(B), int(n_), int(d_), C.c_double(v_))
    return BSR(NP, NP, indexL, itemL, indexU, itemU, D, AL[:9 * itemL.size], AU[:9 * itemU.size], B)


def update_linear(elemopt, coord, conn, E, nu, disp, elem_mat=None):
    """fstr_UpdateNewton of a linear static analysis (orc_update_linear): strain / stress (n_elem, 8, 6) and QFORCE (3 * n_node)
    from the total displacement.  E, nu: scalars or per-material arrays with elem_mat (1-based)."""
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    disp = np.ascontiguousarray(disp, dtype=np.float64)
    Es = np.atleast_1d(np.asarray(E, dtype=np.float64)).copy()
    nus = np.atleast_1d(np.asarray(nu, dtype=np.float64)).copy()
    em = None if elem_mat is None else np.ascontiguousarray(elem_mat, dtype=np.int32)
    ne, nn = conn.shape[0], coord.shape[0]
    strain, stress, qf = np.zeros((ne, 8, 6)), np.zeros((ne, 8, 6)), np.zeros(3 * nn)
    lib().orc_update_linear(int(elemopt), nn, ne, _dp(coord), _ip(conn), _dp(Es), _dp(nus), None if em is None else _ip(em),
                            _dp(disp), _dp(strain), _dp(stress), _dp(qf))
    return strain, stress, qf


# ---------------------------------------------------------------- nonlinear (elastoplastic) path
class _Material(C.Structure):
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("plastic", C.c_int32), ("harden", C.c_int32),
                ("nlgeom", C.c_int32), ("ntab", C.c_int32), ("plconst", C.c_double * 3), ("tab", C.c_void_p)]


class _GaussState(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("stress", "strain", "stress_bak", "strain_bak", "plstrain", "fstat", "istat")]


def cmaterial(mat):
    """mat: refrun.Material-like (E, nu, plastic, harden, plconst, table, nlgeom)."""
    tab = np.ascontiguousarray(mat.table, dtype=np.float64).reshape(-1, 2)
    s = _Material(mat.E, mat.nu, int(mat.plastic), mat.harden, mat.nlgeom, tab.shape[0],
                  (C.c_double * 3)(*mat.plconst), _p(tab) if tab.size else None)
    s._keep = tab
    return s


def new_state(n_elem):
    """tGaussStatus after fstr_init_gauss (mechgauss.f90:37-71), flat arrays."""
    st = {k: np.zeros((n_elem, 8, 6)) for k in ("stress", "strain", "stress_bak", "strain_bak")}
    st["plstrain"] = np.zeros((n_elem, 8))
    st["fstat"] = np.zeros((n_elem, 8))
    st["istat"] = np.zeros((n_elem, 8), dtype=np.int32)
    return st


def cstate(st):
    s = _GaussState(*[_p(st[k]) for k in ("stress", "strain", "stress_bak", "strain_bak", "plstrain", "fstat", "istat")])
    s._keep = st
    return s


def nl_reset_latch():
    lib().orc_nl_reset_latch()


def curr_yield(mat, p):
    lib().orc_curr_yield.restype = C.c_double
    cm = cmaterial(mat)
    return lib().orc_curr_yield(C.byref(cm), C.c_double(p))


def harden_coeff(mat, p):
    lib().orc_harden_coeff.restype = C.c_double
    cm = cmaterial(mat)
    return lib().orc_harden_coeff(C.byref(cm), C.c_double(p))


def elastoplastic_matrix(mat, stress, istat, extval1):
    cm = cmaterial(mat)
    D = np.zeros((6, 6))
    s = np.ascontiguousarray(stress, dtype=np.float64)
    lib().orc_elastoplastic_matrix(C.byref(cm), _dp(s), int(istat), C.c_double(extval1), _dp(D))
    return D


def backward_euler(mat, stress, plstrain, istat, fstat1):
    cm = cmaterial(mat)
    s = np.ascontiguousarray(stress, dtype=np.float64).copy()
    ist, fs = C.c_int32(int(istat)), C.c_double(float(fstat1))
    lib().orc_backward_euler(C.byref(cm), _dp(s), C.c_double(plstrain), C.byref(ist), C.byref(fs))
    return s, ist.value, fs.value


def nl_elements(mat, coord, conn, unode, dunode, state):
    """STF (u+du) -> Update -> STF again, element by element, as oracle/ref_nl_driver.f90 mode 1.
    The latch is reset first (a fresh process).  Returns ke_before, qf, ke_after, new state."""
    nl_reset_latch()
    cm = cmaterial(mat)
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    ne = conn.shape[0]
    st = {k: np.ascontiguousarray(v).copy() for k, v in state.items()}
    ke0, ke1, qf = np.zeros((ne, 24, 24)), np.zeros((ne, 24, 24)), np.zeros((ne, 24))
    U = np.asarray(unode).reshape(-1, 3)
    DU = np.asarray(dunode).reshape(-1, 3)

    def stf(out):
        for e in range(ne):
            nd = conn[e] - 1
            ec = np.ascontiguousarray(coord[nd]); ut = np.ascontiguousarray(U[nd] + DU[nd])
            lib().orc_stf_c3d8bbar_nl(C.byref(cm), _dp(ec), _dp(ut), _dp(st["stress"][e]), _ip(st["istat"][e]),
                                      _dp(st["fstat"][e]), _dp(out[e]))
    stf(ke0)
    for e in range(ne):
        nd = conn[e] - 1
        ec = np.ascontiguousarray(coord[nd]); u = np.ascontiguousarray(U[nd]); du = np.ascontiguousarray(DU[nd])
        lib().orc_update_c3d8bbar(C.byref(cm), _dp(ec), _dp(u), _dp(du), _dp(st["stress"][e]), _dp(st["strain"][e]),
                                  _dp(st["stress_bak"][e]), _dp(st["strain_bak"][e]), _dp(st["plstrain"][e]),
                                  _ip(st["istat"][e]), _dp(st["fstat"][e]), _dp(qf[e]))
    stf(ke1)
    return ke0, qf, ke1, st


class NonlinearModel:
    """Mesh-level restatement: fstr_StiffMatrix + fstr_AddBC, fstr_UpdateNewton + fstr_Update_NDForce,
    fstr_UpdateState, and the fstr_Newton / fstr_solve_NLGEOM control flow around them."""

    def __init__(self, mat, coord, conn, elem_mat=None):
        """mat: one material, or a list of materials with elem_mat (1-based material id per element: several sections)."""
        from .refrun import BSR
        self.mats = list(mat) if isinstance(mat, (list, tuple)) else None
        if self.mats is not None:
            self.elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
            self._cms = [cmaterial(x) for x in self.mats]
            self._cmarr = (_Material * len(self._cms))(*self._cms)
            mat = self.mats[0]
        self.mat, self.cm = mat, cmaterial(mat)
        self.coord = np.ascontiguousarray(coord, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.NP, self.ne = self.coord.shape[0], self.conn.shape[0]
        iL, tL, iU, tU = mat_con(self.NP, self.conn)
        self.m = BSR(self.NP, self.NP, iL, tL, iU, tU, np.zeros(9 * self.NP), np.zeros(9 * max(tL.size, 1)),
                     np.zeros(9 * max(tU.size, 1)))
        self.state = new_state(self.ne)
        self.unode = np.zeros(3 * self.NP)
        self.dunode = np.zeros(3 * self.NP)
        self.qforce = np.zeros(3 * self.NP)
        nl_reset_latch()

    def _sections(self):
        if self.mats is not None:
            lib().orc_nl_set_sections(self._cmarr, _ip(self.elem_mat))
        else:
            lib().orc_nl_set_sections(None, None)

    def stiffness(self):
        self._sections()
        m, cs = self.m, cstate(self.state)
        lib().orc_nl_stiffness(C.byref(self.cm), self.NP, self.ne, _dp(self.coord), _ip(self.conn), _dp(self.unode),
                               _dp(self.dunode), C.byref(cs), _ip(m.indexL), _ip(m.itemL), _ip(m.indexU),
                               _ip(m.itemU), _dp(m.D), _dp(m.AL), _dp(m.AU))

    def add_bc(self, node, dof, val):
        m = self.m
        for n_, d_, v_ in zip(node, dof, val):
            lib().orc_mat_ass_bc(self.NP, _ip(m.indexL), _ip(m.itemL), _ip(m.indexU), _ip(m.itemU), _dp(m.D),
                                 _dp(m.AL), _dp(m.AU), _dp(m.B), int(n_), int(d_), C.c_double(v_))

    def update(self):
        self._sections()
        cs = cstate(self.state)
        lib().orc_nl_update(C.byref(self.cm), self.NP, self.ne, _dp(self.coord), _ip(self.conn), _dp(self.unode),
                            _dp(self.dunode), C.byref(cs), _dp(self.qforce))

    def commit(self):
        self._sections()
        cs = cstate(self.state)
        lib().orc_nl_commit(C.byref(self.cm), self.ne, C.byref(cs))

    def run_steps(self, bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R, nthreads=2, factors=None):
        """fstr_solve_NLGEOM.f90:100-121 + fstr_Newton (fstr_solve_NonLinear.f90:29-167).  factors: (FACTOR(1),
        FACTOR(2)) of a single substep instead of the linear ramp over nsub substeps."""
        log = []
        I = np.ascontiguousarray(I, dtype=np.int32).copy()
        bc_idx = 3 * (np.asarray(bc_node, dtype=np.int64) - 1) + np.asarray(bc_dof, dtype=np.int64) - 1
        # the preconditioner lives across the solves of a run and is refreshed by the recycle policy only
        # (hecmw_matrix_misc.f90:678-697: Iarray(97) = 1 re-uses it up to maxrecycle = 3 times)
        lib().orc_persist_precond(2 if getattr(self, "_persist", False) else 1)
        self._persist = True
        try:
            return self._run_steps(bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R, nthreads, factors, log, bc_idx)
        finally:
            lib().orc_persist_precond(3)

    def _run_steps(self, bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R, nthreads, factors, log, bc_idx):
        for sub in range(1, nsub + 1):
            f1, f2 = ((sub - 1) / nsub, sub / nsub) if factors is None else factors
            self.dunode[:] = 0.0
            GL = cload * f2
            self.m.B[:] = GL - self.qforce
            for it in range(1, max_iter + 1):
                self.stiffness()
                self.add_bc(bc_node, bc_dof, bc_val * (f2 - f1) if it == 1 else np.zeros(len(bc_val)))
                I[96] = 2 if it == 1 else 1
                self.m.X[:] = 0.0
                r = solve_iterative(self.m, I, R, nthreads=nthreads)
                I = r["Iarray"]
                X = r["X"]
                self.dunode += X
                self.update()
                self.m.B[:] = GL - self.qforce
                self.m.B[bc_idx] = 0.0
                res = np.sqrt(np.dot(self.m.B, self.m.B))
                xnrm = np.sqrt(np.dot(X, X))
                qnrm = np.sqrt(np.dot(self.qforce, self.qforce))
                if qnrm < 1e-8:
                    qnrm = 1.0
                dunrm = xnrm if it == 1 else np.sqrt(np.dot(self.dunode, self.dunode))
                log.append((sub, it, r["iter"], res, xnrm, qnrm, dunrm))
                if I[80] == 1 and (res / qnrm < converg or xnrm / dunrm < converg):
                    break
            self.unode += self.dunode
            self.commit()
        return np.array(log)
