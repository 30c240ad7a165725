"""Host-side mirror of the reference's nonlinear static loop on top of the C ABI
(include/fistr_hip.h, "nonlinear static loop" section).  Same names and argument meaning as

    tMaterial            fistr1/src/lib/physics/material.f90:135-149  (+ !ELASTIC / !PLASTIC cards,
                         fistr1/src/common/fstr_ctrl_material.f90:60-106, :341-480)
    fstr_solid           fistr1/src/lib/m_fstr.f90 (unode, dunode, QFORCE, elements(:)%gausses(:))
    fstr_StiffMatrix     fistr1/src/analysis/static/fstr_StiffMatrix.f90:18
    fstr_AddBC           fistr1/src/analysis/static/fstr_AddBC.f90:17
    fstr_UpdateNewton    fistr1/src/analysis/static/fstr_Update.f90:25
    fstr_Update_NDForce  fistr1/src/analysis/static/fstr_Residual.f90:23
    fstr_UpdateState     fistr1/src/analysis/static/fstr_Update.f90:296
    fstr_cutback_save / _load  fistr1/src/analysis/static/fstr_Cutback.f90:108-198
    fstr_Newton          fistr1/src/analysis/static/fstr_solve_NonLinear.f90:29
    fstr_solve_NLGEOM    fistr1/src/analysis/static/fstr_solve_NLGEOM.f90:32 (sub-step loop, linear load ramp)

for one TYPE=361 B-bar group with one isotropic (Mises elastoplastic or elastic) material per section.  Everything is
resident on the GPU; there is NO CPU fallback.
"""
import ctypes as C

import numpy as np

from . import hecmw
from .hecmw import _chk, _ptr, lib

INFINITE, TOTALLAG, UPDATELAG = 0, 1, 2
BILINEAR, MULTILINEAR, SWIFT, RAMBERG_OSGOOD = 0, 1, 2, 3


class _MaterialView(C.Structure):
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("plastic", C.c_int32), ("harden", C.c_int32),
                ("nlgeom", C.c_int32), ("ntab", C.c_int32), ("plconst", C.c_double * 3), ("tab", C.c_void_p)]


class _StateView(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("stress", "strain", "stress_bak", "strain_bak", "plstrain", "fstat",
                                          "istat", "unode", "dunode", "qforce")] + [("latch", C.c_int32)]


class tMaterial:
    """!ELASTIC E, nu; optional !PLASTIC, YIELD=MISES, HARDEN=<harden> with `plconst` (BILINEAR: yield0, H;
    SWIFT / RAMBERG-OSGOOD: the three constants) or `table` rows (yield stress, plastic strain) for MULTILINEAR.
    nlgeom_flag defaults to UPDATELAG as !PLASTIC does (KIRCHHOFF -> TOTALLAG, INFINITE -> INFINITE)."""

    def __init__(self, E, nu, plastic=False, harden=BILINEAR, plconst=(0.0, 0.0, 0.0), table=None, nlgeom_flag=UPDATELAG):
        self.E, self.nu, self.plastic, self.harden = float(E), float(nu), bool(plastic), int(harden)
        self.plconst = tuple(float(v) for v in plconst)
        self.table = np.zeros((0, 2)) if table is None else np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 2)
        self.nlgeom_flag = int(nlgeom_flag)
        if self.plastic and self.harden == MULTILINEAR:
            if self.table.shape[0] < 1 or self.table[0, 1] != 0.0:
                raise ValueError("Multilinear hardening: First plastic strain must be zero")   # fstr_ctrl_material.f90:416
            if (self.table[:, 1] < 0).any():
                raise ValueError("Multilinear hardening: Error in plastic strain definition")

    def view(self):
        v = _MaterialView(self.E, self.nu, int(self.plastic), self.harden, self.nlgeom_flag, self.table.shape[0],
                          (C.c_double * 3)(*self.plconst), _ptr(self.table) if self.table.size else None)
        v._keep = self.table
        return v


class fstr_solid:
    """The resident nonlinear state of one context (created by fx_nl_init)."""

    def __init__(self, ctx, hecMESH_coord, hecMESH_conn, material, elem_mat=None):
        """material: one tMaterial, or a list of them with elem_mat (1-based material id per element = the section's
        material, hecMESH%section_ID -> fstrSOLID%materials)."""
        self.ctx = ctx
        self.coord = np.ascontiguousarray(hecMESH_coord, dtype=np.float64)
        self.conn = np.ascontiguousarray(hecMESH_conn, dtype=np.int32)
        self.material = material
        self.n_node, self.n_elem = self.coord.shape[0], self.conn.shape[0]
        mv = hecmw._MeshView(self.n_node, self.n_elem, _ptr(self.coord), _ptr(self.conn))
        if isinstance(material, (list, tuple)):
            views = [m.view() for m in material]
            arr = (_MaterialView * len(views))(*views)
            self.elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
            if self.elem_mat.shape != (self.n_elem,):
                raise ValueError("elem_mat: one material id per element")
            _chk(lib().fx_nl_init_sections(ctx.h, C.byref(mv), len(views), arr, _ptr(self.elem_mat)))
            return
        m = material.view()
        _chk(lib().fx_nl_init(ctx.h, C.byref(mv), C.byref(m)))

    # ---- state transfer (tests, restart, output)
    def get_state(self, names=("stress", "strain", "stress_bak", "strain_bak", "plstrain", "fstat", "istat",
                               "unode", "dunode", "qforce")):
        ne, nn = self.n_elem, self.n_node
        shapes = {"stress": (ne, 8, 6), "strain": (ne, 8, 6), "stress_bak": (ne, 8, 6), "strain_bak": (ne, 8, 6),
                  "plstrain": (ne, 8), "fstat": (ne, 8), "istat": (ne, 8), "unode": (3 * nn,), "dunode": (3 * nn,),
                  "qforce": (3 * nn,)}
        out = {k: np.zeros(shapes[k], dtype=np.int32 if k == "istat" else np.float64) for k in names}
        v = _StateView(*[_ptr(out.get(k)) for k in shapes], 0)
        _chk(lib().fx_nl_get_state(self.ctx.h, C.byref(v)))
        out["latch"] = int(v.latch)
        return out

    def set_state(self, state, latch=-1):
        keys = ("stress", "strain", "stress_bak", "strain_bak", "plstrain", "fstat", "istat", "unode", "dunode", "qforce")
        arrs = {}
        for k in keys:
            if k in state and state[k] is not None:
                arrs[k] = np.ascontiguousarray(state[k], dtype=np.int32 if k == "istat" else np.float64)
        v = _StateView(*[_ptr(arrs.get(k)) for k in keys], int(latch))
        _chk(lib().fx_nl_set_state(self.ctx.h, C.byref(v)))

    def element_tangents(self):
        ke = np.zeros((self.n_elem, 24, 24))
        _chk(lib().fx_nl_element_tangents(self.ctx.h, _ptr(ke)))
        return ke

    def element_update(self):
        qf = np.zeros((self.n_elem, 24))
        _chk(lib().fx_nl_element_update(self.ctx.h, _ptr(qf)))
        return qf


def _bc_arrays(bc):
    if bc is None:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0)
    return (np.ascontiguousarray(bc[0], dtype=np.int32), np.ascontiguousarray(bc[1], dtype=np.int32),
            np.ascontiguousarray(bc[2], dtype=np.float64))


def fstr_StiffMatrix(fstrSOLID, bc=None):
    """fstr_StiffMatrix + fstr_AddBC: tangent of the current state into the resident matrix, Dirichlet
    elimination with the increments bc = (node, dof, value).  Returns the kernel time in ms."""
    bn, bd, bv = _bc_arrays(bc)
    ms = C.c_float(0)
    _chk(lib().fx_nl_stiffness(fstrSOLID.ctx.h, int(bn.size), _ptr(bn), _ptr(bd), _ptr(bv), C.byref(ms)))
    return ms.value


def fstr_UpdateNewton(fstrSOLID):
    """dunode += X; fstr_UpdateNewton; fstr_Update_NDForce.  Returns (res, xnrm, qnrm, dunrm_all) and the
    kernel time in ms."""
    out = (C.c_double * 4)()
    ms = C.c_float(0)
    _chk(lib().fx_nl_update(fstrSOLID.ctx.h, out, C.byref(ms)))
    return tuple(float(np.sqrt(v)) for v in out), ms.value


def fstr_UpdateState(fstrSOLID):
    _chk(lib().fx_nl_commit(fstrSOLID.ctx.h))


def fstr_cutback_save(fstrSOLID):
    """fstr_Cutback.f90:108-152 for the device-resident quadrature-point history."""
    _chk(lib().fx_nl_snapshot(fstrSOLID.ctx.h, 0))


def fstr_cutback_load(fstrSOLID):
    """fstr_Cutback.f90:155-198: back to the state of the last fstr_cutback_save."""
    _chk(lib().fx_nl_snapshot(fstrSOLID.ctx.h, 1))


def fstr_Newton(fstrSOLID, hecMAT, factor, bc, cload, max_iter, converg, commit_unconverged=False):
    """One substep of fstr_Newton.  factor = (FACTOR(1), FACTOR(2)); bc / cload are the values at load factor 1.
    Returns (converged, log) with log rows (iter, solver iterations, solver code, |B|, |X|, |QFORCE|, |dunode|)."""
    bn, bd, bv = _bc_arrays(bc)
    cl = None if cload is None else np.ascontiguousarray(cload, dtype=np.float64)
    log = np.zeros((max_iter, 7))
    nit = C.c_int32(0)
    code = lib().fx_newton_substep(fstrSOLID.ctx.h, C.c_double(factor[0]), C.c_double(factor[1]), int(bn.size), _ptr(bn),
                                   _ptr(bd), _ptr(bv), _ptr(cl), int(max_iter), C.c_double(converg), _ptr(hecMAT.Iarray),
                                   _ptr(hecMAT.Rarray), _ptr(log), C.byref(nit), int(commit_unconverged))
    _chk(code, allow=(hecmw.HECMW_SOLVER_ERROR_NOCONV_MAXIT, FX_NEWTON_MAXRES))
    fstrSOLID.last_code = code
    return code == 0, log[:nit.value].copy()


FX_NEWTON_MAXRES = 4002


def fstr_set_step_control(fstrSOLID, maxres=1.0e10, is_linear=False):
    """step_ctrl(cstep)%maxres (m_step.f90:31, :78) and fstr_Newton's isLinear (.not. fstrPR%nlgeom)."""
    _chk(lib().fx_nl_set_step_control(fstrSOLID.ctx.h, C.c_double(maxres), int(is_linear)))


def fstr_solve_NLGEOM(fstrSOLID, hecMAT, bc, cload, substeps, max_iter, converg, commit_unconverged=True):
    """The sub-step loop of fstr_solve_NLGEOM with the default linear load-factor ramp (table_nlsta without
    amplitude).  Returns the concatenated Newton log with the substep number in front."""
    logs = []
    for sub in range(1, substeps + 1):
        ok, log = fstr_Newton(fstrSOLID, hecMAT, ((sub - 1) / substeps, sub / substeps), bc, cload, max_iter, converg,
                              commit_unconverged)
        logs.append(np.concatenate([np.full((log.shape[0], 1), float(sub)), log], axis=1))
    return np.concatenate(logs, axis=0)
