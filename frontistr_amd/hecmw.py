"""Host-side mirror of the reference's solver interface on top of the C ABI
(include/fistr_hip.h -> frontistr_amd/libfistr_hip.so).

The reference's host language is Fortran; the production binding is the shim
frontistr_amd/shim/hecmw_solver_hip.f90.  This module is
the same boundary for Python callers, tests and bench.py, with the reference's
names and argument meaning:

    hecmwST_matrix       hecmw1/src/common/hecmw_util_f.F90:433-468
    hecmwST_local_mesh   (communication part) hecmw_util_f.F90:298-310
    hecmw_mat_init       hecmw1/src/solver/matrix/hecmw_matrix_misc.f90:142-182
    hecmw_mat_con        hecmw1/src/solver/matrix/hecmw_mat_con.f90:23
    hecmw_solve          hecmw1/src/solver/hecmw_solver.f90:9
    hecmw_matvec         hecmw1/src/solver/las/hecmw_solver_las.f90:57
    fstr_StiffMatrix     fistr1/src/analysis/static/fstr_StiffMatrix.f90:18 (+ hecmw_mat_ass_bc)

There is NO CPU fallback: a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "libfistr_hip.so")
_lib = None

# hecmw_solve_error.f90:9-15
HECMW_SOLVER_ERROR_INCONS_PC = 1001
HECMW_SOLVER_ERROR_ZERO_DIAG = 2001
HECMW_SOLVER_ERROR_ZERO_RHS = 2002
HECMW_SOLVER_ERROR_NOCONV_MAXIT = 3001
HECMW_SOLVER_ERROR_DIVERGE_MAT = 3002
HECMW_SOLVER_ERROR_DIVERGE_PC = 3003

FX_UP_PROFILE, FX_UP_VALUES, FX_UP_RHS, FX_UP_X, FX_UP_ALL = 1, 2, 4, 8, 15


class HecmwSolverError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("HEC-MW-SOLVER %d: %s" % (code, msg))
        self.code = code


class _MatrixView(C.Structure):
    _fields_ = [("N", C.c_int32), ("NP", C.c_int32), ("NPL", C.c_int32), ("NPU", C.c_int32), ("NDOF", C.c_int32),
                ("indexL", C.c_void_p), ("itemL", C.c_void_p), ("indexU", C.c_void_p), ("itemU", C.c_void_p),
                ("D", C.c_void_p), ("AL", C.c_void_p), ("AU", C.c_void_p), ("B", C.c_void_p), ("X", C.c_void_p)]


class _CommView(C.Structure):
    _fields_ = [("my_rank", C.c_int32), ("PETOT", C.c_int32), ("nn_internal", C.c_int32), ("n_node", C.c_int32),
                ("n_neighbor_pe", C.c_int32),
                ("neighbor_pe", C.c_void_p), ("import_index", C.c_void_p), ("import_item", C.c_void_p),
                ("export_index", C.c_void_p), ("export_item", C.c_void_p)]


class _SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("method", C.c_int32), ("precond", C.c_int32), ("ncolor", C.c_int32),
                ("n_hist", C.c_int32), ("resid", C.c_double), ("rel_resid", C.c_double),
                ("time_setup", C.c_double), ("time_sol", C.c_double), ("time_comm", C.c_double),
                ("time_matvec", C.c_double), ("time_precond", C.c_double)]


class _MeshView(C.Structure):
    _fields_ = [("n_node", C.c_int32), ("n_elem", C.c_int32), ("coord", C.c_void_p), ("conn", C.c_void_p)]


def lib():
    """Load libfistr_hip.so (built by `make -C frontistr_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise ImportError("%s missing: run `make -C frontistr_amd/csrc` (hipcc, gfx950). "
                              "There is no CPU fallback." % LIBPATH)
        L = C.CDLL(LIBPATH)
        L.fx_last_error.restype = C.c_char_p
        L.fx_version.restype = C.c_char_p
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(code, allow=()):
    if code == 0 or code in allow:
        return code
    raise HecmwSolverError(code, lib().fx_last_error().decode(errors="replace"))


class hecmwST_matrix:
    """Same members as the reference type; index arrays are int32, items 1-based."""

    def __init__(self):
        self.N = self.NP = self.NPL = self.NPU = 0
        self.NDOF = 3
        self.indexL = self.itemL = self.indexU = self.itemU = None
        self.D = self.AL = self.AU = self.B = self.X = None
        self.Iarray = np.zeros(100, dtype=np.int32)
        self.Rarray = np.zeros(100, dtype=np.float64)
        hecmw_mat_init(self)

    @classmethod
    def from_arrays(cls, N, NP, indexL, itemL, indexU, itemU, D, AL, AU, B=None, X=None, NDOF=3):
        m = cls()
        m.N, m.NP, m.NDOF = int(N), int(NP), int(NDOF)
        m.indexL = np.ascontiguousarray(indexL, dtype=np.int32)
        m.itemL = np.ascontiguousarray(itemL, dtype=np.int32)
        m.indexU = np.ascontiguousarray(indexU, dtype=np.int32)
        m.itemU = np.ascontiguousarray(itemU, dtype=np.int32)
        m.NPL, m.NPU = int(m.itemL.size), int(m.itemU.size)
        m.D = None if D is None else np.ascontiguousarray(D, dtype=np.float64)
        m.AL = None if AL is None else np.ascontiguousarray(AL, dtype=np.float64)
        m.AU = None if AU is None else np.ascontiguousarray(AU, dtype=np.float64)
        m.B = np.zeros(m.NDOF * m.NP) if B is None else np.ascontiguousarray(B, dtype=np.float64)
        m.X = np.zeros(m.NDOF * m.NP) if X is None else np.ascontiguousarray(X, dtype=np.float64)
        return m

    def view(self):
        v = _MatrixView(self.N, self.NP, self.NPL, self.NPU, self.NDOF, _ptr(self.indexL), _ptr(self.itemL),
                        _ptr(self.indexU), _ptr(self.itemU), _ptr(self.D), _ptr(self.AL), _ptr(self.AU),
                        _ptr(self.B), _ptr(self.X))
        v._keep = self
        return v


def hecmw_mat_init(hecMAT):
    """hecmw_matrix_misc.f90:142-182 defaults (1-based slot k lives at index k-1)."""
    I, R = hecMAT.Iarray, hecMAT.Rarray
    I[:] = 0
    R[:] = 0.0
    I[0] = 100      # iter
    I[1] = 1        # method  CG
    I[2] = 1        # precond SSOR
    I[4] = 1        # iterPREmax
    I[5] = 10       # nrest
    I[33] = 10      # ncolor_in
    I[34] = 3       # maxrecycle_precond
    I[12] = 3       # mpc_method
    I[96] = 1       # flag_numfact
    I[97] = 1       # flag_symbfact
    I[98] = 1       # solver_type: iterative
    R[0] = 1.0e-8   # resid
    R[1] = 1.0      # sigma_diag
    R[3] = 0.10     # thresh
    R[4] = 0.10     # filter
    R[10] = 1.0e4   # penalty


class hecmwST_local_mesh:
    """Only what the hot path reads (hecmw_util_f.F90:298-310) + mesh arrays for assembly."""

    def __init__(self, n_node=0, nn_internal=None):
        self.n_node = n_node
        self.nn_internal = n_node if nn_internal is None else nn_internal
        self.my_rank, self.PETOT, self.zero = 0, 1, 0
        self.n_neighbor_pe = 0
        self.neighbor_pe = np.zeros(0, dtype=np.int32)
        self.import_index = np.zeros(1, dtype=np.int32)
        self.import_item = np.zeros(0, dtype=np.int32)
        self.export_index = np.zeros(1, dtype=np.int32)
        self.export_item = np.zeros(0, dtype=np.int32)
        self.node = None            # 3*n_node coordinates
        self.elem_node_item = None  # flattened connectivity, 1-based
        self.n_elem = 0

    def comm_view(self):
        v = _CommView(self.my_rank, self.PETOT, self.nn_internal, self.n_node, self.n_neighbor_pe,
                      _ptr(self.neighbor_pe), _ptr(self.import_index), _ptr(self.import_item),
                      _ptr(self.export_index), _ptr(self.export_item))
        v._keep = self
        return v


class SolverContext:
    """Device state the reference keeps in module-level `save` variables."""

    def __init__(self, device=-1):
        self.h = C.c_void_p()
        _chk(lib().fx_create(device, C.byref(self.h)))
        self.info = None
        self.history = None
        self.messages = []

    def close(self):
        if self.h:
            lib().fx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- staged device-resident API -------------------------------------
    def upload(self, hecMAT, hecMESH=None, what=FX_UP_ALL):
        mv = hecMAT.view()
        cv = hecMESH.comm_view() if hecMESH is not None else None
        _chk(lib().fx_upload(self.h, C.byref(mv), C.byref(cv) if cv is not None else None, what))

    def precond_setup(self, hecMAT):
        _chk(lib().fx_precond_setup(self.h, _ptr(hecMAT.Iarray), _ptr(hecMAT.Rarray)))

    def set_option(self, name, value):
        """Tuning knob of the live context (the FX_* names of csrc/fx_internal.h)."""
        f = lib().fx_set_option
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        _chk(f(self.h, name.encode(), float(value)))

    def placement_report(self):
        """Where the value arrays of the sliced layouts live (fx_placement_report): the context's value arena."""
        out = (C.c_double * 9)()
        _chk(lib().fx_placement_report(self.h, out))
        return {"arena_bytes": int(out[0]), "arena_used_bytes": int(out[1]), "arrays_in_arena": int(out[2]),
                "spmv_values_in_arena": bool(out[3]), "lower_values_in_arena": bool(out[4]), "upper_values_in_arena": bool(out[5]),
                "spmv_value_bytes": int(out[6]), "arenas_timed": int(out[7]), "kept_ms": float(out[8])}

    def march_report(self):
        """The plane march of the level-scheduled sweeps of this context (fx_march_report)."""
        out = (C.c_double * 16)()
        _chk(lib().fx_march_report(self.h, out))
        keys = ("built", "chunk_rows", "chunks", "pair_waves", "rounds_fwd", "rounds_bwd", "near_blocks", "far_blocks",
                "far_same_chunk", "est_us", "est_level_us", "build_s", "applies", "grid", "max_round_rows", "levels")
        d = {k: float(out[i]) for i, k in enumerate(keys)}
        for k in keys:
            if k not in ("est_us", "est_level_us", "build_s"):
                d[k] = int(d[k])
        return d

    def march_trace(self):
        """Diagnostics: one traced march apply (fx_debug_march_trace) -> array [chunks, 8]: forward start / end, backward start / end
        in microseconds, rounds that waited and polls, forward / backward."""
        n = self.march_report()["chunks"]
        out = np.zeros(8 * max(n, 1))
        nch = C.c_int32(0)
        f = lib().fx_debug_march_trace
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _chk(f(self.h, _ptr(out), out.size, C.byref(nch)))
        a = out.reshape(-1, 8)
        a[:, :4] = np.where(a[:, :4] >= 0, a[:, :4] * 0.01, -1.0)
        return a

    def march_rounds(self, chunk):
        """Diagnostics: when each round of one chunk's forward sweep passed its barrier (us; negative = the round waited)."""
        out = np.zeros(1 << 16)
        n = C.c_int32(0)
        f = lib().fx_debug_march_rounds
        f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        _chk(f(self.h, int(chunk), _ptr(out), out.size, C.byref(n)))
        return out[:n.value].copy()

    def solve_resident(self, hecMAT, want_history=True):
        info = _SolveInfo()
        maxit = int(hecMAT.Iarray[0])
        hist = np.zeros(max(maxit, 1) + 1) if want_history else None   # GMRES logs MAXIT+1 lines when it runs out
        code = lib().fx_solve_resident(self.h, _ptr(hecMAT.Iarray), _ptr(hecMAT.Rarray), C.byref(info),
                                       _ptr(hist), 0 if hist is None else hist.size)
        self._finish(code, info, hist)
        return code

    def download_x(self, hecMAT):
        _chk(lib().fx_download_x(self.h, _ptr(hecMAT.X), hecMAT.X.size))
        return hecMAT.X

    def download_matrix(self, hecMAT):
        for k, n in (("D", 9 * hecMAT.NP), ("AL", 9 * hecMAT.NPL), ("AU", 9 * hecMAT.NPU)):
            if getattr(hecMAT, k) is None:
                setattr(hecMAT, k, np.zeros(max(n, 1)))
        _chk(lib().fx_download_matrix(self.h, _ptr(hecMAT.D), _ptr(hecMAT.AL), _ptr(hecMAT.AU), _ptr(hecMAT.B)))

    def matvec_resident_ms(self, nrepeat=10):
        ms = C.c_float(0)
        _chk(lib().fx_matvec_resident(self.h, nrepeat, C.byref(ms)))
        return ms.value

    def spmv_resident_ms(self, variant=1, nrepeat=10):
        """variant 0 plain, 1 with the fused x.y partial (the CG loop's launch), 2 residual + r.r partial."""
        ms = C.c_float(0)
        _chk(lib().fx_spmv_resident(self.h, int(variant), nrepeat, C.byref(ms)))
        return ms.value

    def precond_apply_ms(self, nrepeat=5):
        ms = C.c_float(0)
        _chk(lib().fx_precond_apply_resident(self.h, nrepeat, C.byref(ms)))
        return ms.value

    def stream_ceiling_gbs(self, nrepeat=5):
        out = C.c_double(0)
        _chk(lib().fx_stream_ceiling(self.h, nrepeat, C.byref(out)))
        return out.value

    def stats(self):
        out = (C.c_int64 * 16)()
        _chk(lib().fx_get_stats(self.h, out))
        keys = ("N", "NP", "NPL", "NPU", "M_pairs", "M_blocks", "M_slices", "ncolor", "L_pairs", "L_blocks",
                "U_pairs", "U_blocks", "ssor_slices", "wg_interior", "wg_boundary", "eisenstat")
        st = {k: int(out[i]) for i, k in enumerate(keys)}
        v = st["eisenstat"]
        st["ssor_natural"] = (v >> 1) & 1     # PRECOND = 1 resident as the natural-order (level-scheduled) SSOR
        st["df_mode"] = (v >> 2) & 3            # 0: one launch per colour / level; 1: dataflow ILU(0) sweeps; 2: also SSOR
        st["df_fallbacks"] = (v >> 8) & 0xFF    # timed-out dataflow sweeps that made the context fall back to df_mode 0
        st["df_grid"] = v >> 16                 # workgroups of the last dataflow launch (after the co-residency clamp)
        st["eisenstat"] = v & 1
        return st

    def krylov_begin(self, hecMAT):
        _chk(lib().fx_krylov_begin(self.h, _ptr(hecMAT.Iarray), _ptr(hecMAT.Rarray)))

    def krylov_steps(self, nsteps):
        it, st, rs = C.c_int32(0), C.c_int32(0), C.c_double(0)
        _chk(lib().fx_krylov_steps(self.h, int(nsteps), C.byref(it), C.byref(st), C.byref(rs)))
        return it.value, st.value, rs.value

    def comm_ledger(self):
        """What this rank asked of the transport since it was set up (fx_comm_ledger): dict with ops, seq_hash, allreduces,
        allreduce_bytes, halos, own_halo_comm and peers = {rank: (sends, send_bytes, recvs, recv_bytes)}."""
        n = C.c_int32(0)
        _chk(lib().fx_comm_ledger(self.h, None, 0, C.byref(n)))
        out = (C.c_int64 * max(n.value, 1))()
        _chk(lib().fx_comm_ledger(self.h, out, n.value, C.byref(n)))
        v = list(out)[:n.value]
        peers = {int(v[7 + 5 * k]): tuple(int(x) for x in v[8 + 5 * k:12 + 5 * k]) for k in range(int(v[5]))}
        return {"ops": int(v[0]), "seq_hash": int(v[1]) & 0xFFFFFFFFFFFFFFFF, "allreduces": int(v[2]), "allreduce_bytes": int(v[3]),
                "halos": int(v[4]), "own_halo_comm": bool(v[6]), "peers": peers}

    def krylov_history(self):
        """RESID per iteration of the staged loop since krylov_begin (the reference's ITERLOG lines, hecmw_solver_CG.f90:245)."""
        n = C.c_int32(0)
        _chk(lib().fx_krylov_history(self.h, None, 0, C.byref(n)))
        h = np.zeros(max(n.value, 1))
        _chk(lib().fx_krylov_history(self.h, _ptr(h), n.value, C.byref(n)))
        return h[:n.value]

    def precond_apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _chk(lib().fx_precond_apply_host(self.h, _ptr(r), _ptr(z)))
        return z

    def dot(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = C.c_double(0)
        _chk(lib().fx_dot_host(self.h, _ptr(x), _ptr(y), C.byref(out)))
        return out.value

    def assemble_c3d8(self, coord, conn, E, nu, elemopt=1, load=None, bc=None, sections=None):
        """fstr_StiffMatrix + fstr_AddBC on the device.  sections = (E[], nu[], elem_mat[] 1-based): several materials."""
        coord = np.ascontiguousarray(coord, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        mv = _MeshView(coord.shape[0], conn.shape[0], _ptr(coord), _ptr(conn))
        if bc is None:
            bn = np.zeros(0, dtype=np.int32); bd = np.zeros(0, dtype=np.int32); bv = np.zeros(0)
        else:
            bn = np.ascontiguousarray(bc[0], dtype=np.int32)
            bd = np.ascontiguousarray(bc[1], dtype=np.int32)
            bv = np.ascontiguousarray(bc[2], dtype=np.float64)
        load = None if load is None else np.ascontiguousarray(load, dtype=np.float64)
        ms = C.c_float(0)
        if sections is not None:
            Es = np.ascontiguousarray(sections[0], dtype=np.float64)
            nus = np.ascontiguousarray(sections[1], dtype=np.float64)
            em = np.ascontiguousarray(sections[2], dtype=np.int32)
            _chk(lib().fx_assemble_c3d8_sections(self.h, C.byref(mv), int(Es.size), _ptr(Es), _ptr(nus), _ptr(em), int(elemopt),
                                                 _ptr(load), int(bn.size), _ptr(bn), _ptr(bd), _ptr(bv), C.byref(ms)))
            return ms.value
        _chk(lib().fx_assemble_c3d8(self.h, C.byref(mv), C.c_double(E), C.c_double(nu), int(elemopt), _ptr(load),
                                    int(bn.size), _ptr(bn), _ptr(bd), _ptr(bv), C.byref(ms)))
        return ms.value

    def update_c3d8_linear(self, coord, conn, E, nu, disp, elemopt=1, elem_mat=None):
        """fstr_UpdateNewton of a linear static analysis on the device (fx_update_c3d8_linear): strain, stress (n_elem, 8, 6) at the
        quadrature points and QFORCE (3 * n_node) from the total displacement; E, nu scalars or per-material arrays with
        elem_mat (1-based).  Returns (strain, stress, qforce, kernel ms)."""
        coord = np.ascontiguousarray(coord, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        Es = np.atleast_1d(np.asarray(E, dtype=np.float64)).copy()
        nus = np.atleast_1d(np.asarray(nu, dtype=np.float64)).copy()
        em = None if elem_mat is None else np.ascontiguousarray(elem_mat, dtype=np.int32)
        mv = _MeshView(coord.shape[0], conn.shape[0], _ptr(coord), _ptr(conn))
        ps, pt = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        qf = np.zeros(3 * coord.shape[0])
        ms = C.c_float(0)
        _chk(lib().fx_update_c3d8_linear(self.h, C.byref(mv), int(Es.size), _ptr(Es), _ptr(nus), _ptr(em), int(elemopt), _ptr(disp),
                                         C.byref(ps), C.byref(pt), _ptr(qf), C.byref(ms)))
        n = 48 * conn.shape[0]
        strain = np.ctypeslib.as_array(ps, shape=(n,)).reshape(-1, 8, 6).copy()      # the library's pinned staging: copy out
        stress = np.ctypeslib.as_array(pt, shape=(n,)).reshape(-1, 8, 6).copy()
        return strain, stress, qf, ms.value

    def element_stiffness(self, elemopt, ecoord, E, nu):
        ec = np.ascontiguousarray(ecoord, dtype=np.float64).reshape(8, 3)
        k = np.zeros((24, 24))
        _chk(lib().fx_element_stiffness_c3d8(self.h, int(elemopt), _ptr(ec), C.c_double(E), C.c_double(nu), _ptr(k)))
        return k

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _chk(lib().fx_comm_init(self.h, buf, rank, nranks))

    def comm_size(self):
        """(ranks, device) as the transport itself reports them (ncclCommCount / ncclCommCuDevice)."""
        n, d = C.c_int32(0), C.c_int32(0)
        _chk(lib().fx_comm_size(self.h, C.byref(n), C.byref(d)))
        return n.value, d.value

    def synchronize(self):
        _chk(lib().fx_device_synchronize(self.h))

    def _finish(self, code, info, hist):
        self.info = info
        self.history = None if hist is None else hist[:info.n_hist].copy()
        if code < 0 or code in (HECMW_SOLVER_ERROR_INCONS_PC, HECMW_SOLVER_ERROR_ZERO_DIAG):
            # E-codes abort in the reference (hecmw_solve_error.f90:50-83)
            raise HecmwSolverError(code, lib().fx_last_error().decode(errors="replace"))
        if code:
            self.messages.append("#### HEC-MW-SOLVER-W-%d" % code)


def comm_unique_id():
    buf = (C.c_ubyte * 128)()
    _chk(lib().fx_comm_unique_id(buf))
    return bytes(buf)


_default_ctx = None


def _ctx(ctx):
    global _default_ctx
    if ctx is not None:
        return ctx
    if _default_ctx is None:
        _default_ctx = SolverContext()
    return _default_ctx


def hecmw_mat_con(hecMESH, hecMAT):
    """CRS block profile from hecMESH%elem_node_item (TYPE=361: 8 nodes per element)."""
    conn = np.ascontiguousarray(hecMESH.elem_node_item, dtype=np.int32).reshape(-1, 8)
    NP = hecMESH.n_node
    indexL = np.zeros(NP + 1, dtype=np.int32)
    indexU = np.zeros(NP + 1, dtype=np.int32)
    _chk(lib().fx_mat_con(NP, conn.shape[0], 8, _ptr(conn), _ptr(indexL), _ptr(indexU), None, None))
    itemL = np.zeros(max(int(indexL[NP]), 1), dtype=np.int32)
    itemU = np.zeros(max(int(indexU[NP]), 1), dtype=np.int32)
    _chk(lib().fx_mat_con(NP, conn.shape[0], 8, _ptr(conn), _ptr(indexL), _ptr(indexU), _ptr(itemL), _ptr(itemU)))
    hecMAT.N, hecMAT.NP = hecMESH.nn_internal, NP
    hecMAT.indexL, hecMAT.indexU = indexL, indexU
    hecMAT.itemL, hecMAT.itemU = itemL[:indexL[NP]], itemU[:indexU[NP]]       # exact-size arrays (views when nothing is cut)
    hecMAT.NPL, hecMAT.NPU = int(indexL[NP]), int(indexU[NP])
    hecMAT.B = np.zeros(3 * NP)
    hecMAT.X = np.zeros(3 * NP)
    return hecMAT


def march_plan(hecMAT, chunk, waves):
    """Host-only: the two plane-march programs of hecMAT's profile (fx_march_plan), replayed on the host."""
    out = (C.c_double * 8)()
    f = lib().fx_march_plan
    f.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    iL, jL = np.ascontiguousarray(hecMAT.indexL, dtype=np.int32), np.ascontiguousarray(hecMAT.itemL, dtype=np.int32)
    iU, jU = np.ascontiguousarray(hecMAT.indexU, dtype=np.int32), np.ascontiguousarray(hecMAT.itemU, dtype=np.int32)
    _chk(f(int(hecMAT.N), _ptr(iL), _ptr(jL), _ptr(iU), _ptr(jU), int(chunk), int(waves), out))
    keys = ("admitted", "chunks", "rounds_fwd", "rounds_bwd", "near_blocks", "far_blocks", "max_round_rows", "levels")
    return {k: int(out[i]) for i, k in enumerate(keys)}


def hecmw_solve(hecMESH, hecMAT, ctx=None, want_history=True):
    """subroutine hecmw_solve(hecMESH, hecMAT): solves hecMAT in place (X, Iarray flags).
    Returns the reference's status code (0, or a W-code such as 3001 / 2002)."""
    ctx = _ctx(ctx)
    mv = hecMAT.view()
    cv = hecMESH.comm_view() if hecMESH is not None else None
    info = _SolveInfo()
    maxit = int(hecMAT.Iarray[0])
    hist = np.zeros(max(maxit, 1) + 1) if want_history else None   # GMRES logs MAXIT+1 lines when it runs out
    code = lib().fx_solve(ctx.h, C.byref(mv), C.byref(cv) if cv is not None else None, _ptr(hecMAT.Iarray),
                          _ptr(hecMAT.Rarray), C.byref(info), _ptr(hist), 0 if hist is None else hist.size)
    ctx._finish(code, info, hist)
    return code


def hecmw_matvec(hecMESH, hecMAT, X, Y, ctx=None):
    """subroutine hecmw_matvec(hecMESH, hecMAT, X, Y, COMMtime): Y(1:3N) = A X; returns COMMtime."""
    ctx = _ctx(ctx)
    mv = hecMAT.view()
    cv = hecMESH.comm_view() if hecMESH is not None else None
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.flags.c_contiguous and Y.flags.c_contiguous
    t = C.c_double(0.0)
    _chk(lib().fx_matvec(ctx.h, C.byref(mv), C.byref(cv) if cv is not None else None, _ptr(X), _ptr(Y), C.byref(t)))
    return t.value
