"""TEST INFRASTRUCTURE ONLY: run the reference executables built by
oracle/build_ref.py (oracle/_ref/ref_solve[_omp], oracle/_ref/ref_fem) on flat
binary files and parse what they print.  Used by tests/, by
tests/golden/make_golden.py and by bench.py's cpu_baseline leg -- never by the
product path.
"""
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "_ref")
MAGIC_SOLVE = 1179210580
MAGIC_FEM = 1179206989


def have_ref(name="ref_solve"):
    return os.path.exists(os.path.join(REFDIR, name))


class BSR:
    """Plain container mirroring hecmwST_matrix's members (1-based items)."""

    def __init__(self, N, NP, indexL, itemL, indexU, itemU, D, AL, AU, B=None, X=None, NDOF=3):
        self.N, self.NP, self.NDOF = int(N), int(NP), int(NDOF)
        self.indexL = np.ascontiguousarray(indexL, dtype=np.int32)
        self.indexU = np.ascontiguousarray(indexU, dtype=np.int32)
        self.itemL = np.ascontiguousarray(itemL, dtype=np.int32)
        self.itemU = np.ascontiguousarray(itemU, dtype=np.int32)
        self.D = np.ascontiguousarray(D, dtype=np.float64)
        self.AL = np.ascontiguousarray(AL, dtype=np.float64)
        self.AU = np.ascontiguousarray(AU, dtype=np.float64)
        self.B = np.zeros(self.NDOF * self.NP) if B is None else np.ascontiguousarray(B, dtype=np.float64)
        self.X = np.zeros(self.NDOF * self.NP) if X is None else np.ascontiguousarray(X, dtype=np.float64)

    @property
    def NPL(self):
        return int(self.itemL.size)

    @property
    def NPU(self):
        return int(self.itemU.size)

    def to_scipy(self):
        import scipy.sparse as sp
        N = self.N
        rows, cols, vals = [], [], []
        for idx, item, A in ((self.indexL, self.itemL, self.AL), (self.indexU, self.itemU, self.AU)):
            nblk = idx[N]
            r = np.repeat(np.arange(N), np.diff(idx[:N + 1]))
            rows.append(r)
            cols.append(item[:nblk] - 1)
            vals.append(A[:9 * nblk].reshape(-1, 3, 3))
        rows.append(np.arange(N))
        cols.append(np.arange(N))
        vals.append(self.D[:9 * N].reshape(-1, 3, 3))
        r = np.concatenate(rows)
        c = np.concatenate(cols)
        v = np.concatenate(vals)
        order = np.lexsort((c, r))
        r, c, v = r[order], c[order], v[order]
        indptr = np.zeros(N + 1, dtype=np.int64)
        np.add.at(indptr, r + 1, 1)
        indptr = np.cumsum(indptr)
        return sp.bsr_matrix((v, c, indptr), shape=(3 * N, 3 * self.NP))


def default_params(method=1, precond=3, maxit=10000, tol=1e-8, iterlog=1, timelog=1,
                   ncolor=10, sigma_diag=1.0, iterpremax=1):
    """Iarray/Rarray as hecmw_mat_init (hecmw_matrix_misc.f90:142-182) + !SOLVER card."""
    I = np.zeros(100, dtype=np.int32)
    R = np.zeros(100, dtype=np.float64)
    I[0] = maxit; I[1] = method; I[2] = precond; I[3] = 0; I[4] = iterpremax
    I[5] = 10; I[6] = 0; I[20] = iterlog; I[21] = timelog
    I[33] = ncolor; I[34] = 3; I[12] = 3
    I[96] = 1; I[97] = 1; I[98] = 1
    R[0] = tol; R[1] = sigma_diag; R[2] = 0.0; R[3] = 0.10; R[4] = 0.10; R[10] = 1.0e4
    return I, R


def write_system(path, mode, m, I, R, nrepeat=1, comm=None):
    with open(path, "wb") as f:
        ndof = int(getattr(m, "NDOF", 3))
        np.array([MAGIC_SOLVE, mode + (100 * ndof if ndof != 3 else 0), m.N, m.NP, m.NPL, m.NPU, nrepeat], dtype=np.int32).tofile(f)
        I.astype(np.int32).tofile(f)
        R.astype(np.float64).tofile(f)
        for a in (m.indexL, m.indexU, m.itemL, m.itemU):
            a.astype(np.int32).tofile(f)
        for a in (m.D, m.AL, m.AU, m.B, m.X):
            a.astype(np.float64).tofile(f)
        if comm is not None:      # halo tables of a subdomain: dict(PETOT, my_rank, neighbor_pe, import_index, export_index, import_item, export_item)
            nb = np.asarray(comm["neighbor_pe"], dtype=np.int32)
            np.array([nb.size, comm.get("PETOT", 1), comm.get("my_rank", 0)], dtype=np.int32).tofile(f)
            for k in ("neighbor_pe", "import_index", "export_index", "import_item", "export_item"):
                np.asarray(comm[k], dtype=np.int32).tofile(f)


HIST_RE = re.compile(r"^\s*(\d+)\s+([0-9.]+E[+-]\d+)\s*$")
HIST3_RE = re.compile(r"^\s*(\d+)\s+(\d+)\s+([0-9.]+E[+-]\d+)\s*$")     # GMRES: iter, I+1, RESID (hecmw_solver_GMRES.f90:272)


def parse_stdout(text):
    hist, info = [], {}
    for line in text.splitlines():
        m = HIST_RE.match(line)
        if m:
            hist.append((int(m.group(1)), float(m.group(2))))
            continue
        m = HIST3_RE.match(line)
        if m:
            hist.append((int(m.group(1)), float(m.group(3))))
            continue
        m = re.match(r"\s*(\d+) iterations\s+([0-9.E+-]+)", line)
        if m:
            info["iter"] = int(m.group(1))
            info["resid"] = float(m.group(2))
            info.setdefault("iters", []).append(int(m.group(1)))      # one entry per solve (mode 4 runs several)
        m = re.match(r"### Relative residual =\s*([0-9.E+-]+)", line)
        if m:
            info["rel_resid"] = float(m.group(1))
        for key, tag in (("setup", "set-up time"), ("solver", "solver time"), ("matvec", "solver/matvec"),
                         ("precond", "solver/precond"), ("per_iter", "solver/1 iter")):
            if tag in line:
                try:
                    info["t_" + key] = float(line.split(":")[1])
                except (IndexError, ValueError):
                    pass
        if line.startswith("### ") and "BLOCK" in line:
            info["banner"] = line.strip()
        if "HEC-MW-SOLVER" in line:
            info.setdefault("messages", []).append(line.strip())
    info["history"] = hist
    return info


def run_solve(m, I, R, mode=1, threads=1, nrepeat=1, workdir=None, timeout=None, exe_name=None, extra_env=None, comm=None):
    """Run the reference.  threads==1 -> serial build (natural-order SSOR);
    threads>=2 -> OpenMP build (RCM + multicolour SSOR)."""
    exe = os.path.join(REFDIR, exe_name or ("ref_solve_omp" if threads > 1 else "ref_solve"))
    if not os.path.exists(exe):
        raise FileNotFoundError(exe + " (run python oracle/build_ref.py where /root/reference exists)")
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        write_system(fin, mode, m, I, R, nrepeat, comm)
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(threads)
        if extra_env:
            env.update(extra_env)
        p = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, env=env, timeout=timeout)
        info = parse_stdout(p.stdout)
        info["stdout"] = p.stdout
        info["returncode"] = p.returncode
        if os.path.exists(fout):
            with open(fout, "rb") as f:
                info["Iarray"] = np.fromfile(f, dtype=np.int32, count=100)
                info["Rarray"] = np.fromfile(f, dtype=np.float64, count=100)
                info["X"] = np.fromfile(f, dtype=np.float64, count=int(getattr(m, "NDOF", 3)) * m.NP)
                info["t_total"] = float(np.fromfile(f, dtype=np.float64, count=1)[0])
    return info


def run_fem(coord, conn, E, nu, bc_node, bc_dof, bc_val, B0, elemopt=1, workdir=None, sections=None):
    """Reference profile + element stiffness + assembly + Dirichlet BC.  sections = (E[], nu[], elem_mat[] 1-based):
    several materials, E / nu arguments ignored."""
    exe = os.path.join(REFDIR, "ref_fem")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    n_node, n_elem = coord.shape[0], conn.shape[0]
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            np.array([MAGIC_FEM, elemopt + (10 if sections is not None else 0), n_node, n_elem, len(bc_node)], dtype=np.int32).tofile(f)
            np.array([E, nu], dtype=np.float64).tofile(f)
            np.ascontiguousarray(coord, dtype=np.float64).tofile(f)
            np.ascontiguousarray(conn, dtype=np.int32).tofile(f)
            np.ascontiguousarray(bc_node, dtype=np.int32).tofile(f)
            np.ascontiguousarray(bc_dof, dtype=np.int32).tofile(f)
            np.ascontiguousarray(bc_val, dtype=np.float64).tofile(f)
            np.ascontiguousarray(B0, dtype=np.float64).tofile(f)
            if sections is not None:
                Es, nus, em = sections
                np.array([len(Es)], dtype=np.int32).tofile(f)
                np.ascontiguousarray(Es, dtype=np.float64).tofile(f)
                np.ascontiguousarray(nus, dtype=np.float64).tofile(f)
                np.ascontiguousarray(em, dtype=np.int32).tofile(f)
        p = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0 or not os.path.exists(fout):
            raise RuntimeError("ref_fem failed: " + p.stdout)
        with open(fout, "rb") as f:
            N, NP, NPL, NPU = np.fromfile(f, dtype=np.int32, count=4)
            indexL = np.fromfile(f, dtype=np.int32, count=NP + 1)
            indexU = np.fromfile(f, dtype=np.int32, count=NP + 1)
            itemL = np.fromfile(f, dtype=np.int32, count=NPL)
            itemU = np.fromfile(f, dtype=np.int32, count=NPU)
            D = np.fromfile(f, dtype=np.float64, count=9 * NP)
            AL = np.fromfile(f, dtype=np.float64, count=9 * NPL)
            AU = np.fromfile(f, dtype=np.float64, count=9 * NPU)
            B = np.fromfile(f, dtype=np.float64, count=3 * NP)
            ke = np.fromfile(f, dtype=np.float64, count=576).reshape(24, 24).T.copy()
            t = float(np.fromfile(f, dtype=np.float64, count=1)[0])
    return BSR(N, NP, indexL, itemL, indexU, itemU, D, AL, AU, B), ke, t


MAGIC_UPD = 1179210064


def run_update(elemopt, coord, conn, E, nu, unode, dunode, elem_mat=None, workdir=None):
    """Reference stress update of a linear static analysis, element by element (oracle/ref_update_driver.f90):
    -> strain, stress (n_elem, 8, 6), qforce (3 * n_node)."""
    exe = os.path.join(REFDIR, "ref_update")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    Es = np.atleast_1d(np.asarray(E, dtype=np.float64))
    nus = np.atleast_1d(np.asarray(nu, dtype=np.float64))
    ne, nn = conn.shape[0], coord.shape[0]
    em = np.ones(ne, dtype=np.int32) if elem_mat is None else np.ascontiguousarray(elem_mat, dtype=np.int32)
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            np.array([MAGIC_UPD, elemopt, nn, ne, Es.size], dtype=np.int32).tofile(f)
            Es.tofile(f); nus.tofile(f); em.tofile(f)
            coord.tofile(f); conn.tofile(f)
            np.ascontiguousarray(unode, dtype=np.float64).tofile(f)
            np.ascontiguousarray(dunode, dtype=np.float64).tofile(f)
        p = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
        if p.returncode != 0 or not os.path.exists(fout):
            raise RuntimeError("ref_update failed: " + p.stdout)
        with open(fout, "rb") as f:
            strain = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            stress = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            qf = np.fromfile(f, dtype=np.float64, count=3 * nn)
    return strain, stress, qf


MAGIC_NL = 1179209292


class Material:
    """What fstr_ctrl_get_ELASTICITY / fstr_ctrl_get_PLASTICITY leave in tMaterial for an isotropic
    Mises material (fstr_ctrl_material.f90:60-106, :341-480).  harden: 0 BILINEAR, 1 MULTILINEAR,
    2 SWIFT, 3 RAMBERG-OSGOOD; table rows = (yield stress, plastic strain); nlgeom: 0 INFINITE,
    1 TOTALLAG (KIRCHHOFF), 2 UPDATELAG (default of !PLASTIC)."""

    def __init__(self, E, nu, plastic=False, harden=0, plconst=(0.0, 0.0, 0.0), table=None, nlgeom=2):
        self.E, self.nu, self.plastic, self.harden = float(E), float(nu), bool(plastic), int(harden)
        self.plconst = tuple(float(v) for v in plconst)
        self.table = np.zeros((0, 2)) if table is None else np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 2)
        self.nlgeom = int(nlgeom)


def _write_nl_head(f, mode, mat, coord, conn, bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R, elem_mat=None):
    """mat: one Material, or a list with elem_mat (1-based ids): mode + 10 tells the driver that further materials follow."""
    mats = list(mat) if isinstance(mat, (list, tuple)) else [mat]
    mat = mats[0]
    if len(mats) > 1:
        mode += 10
    n_node, n_elem = coord.shape[0], conn.shape[0]
    np.array([MAGIC_NL, mode, n_node, n_elem, len(bc_node), nsub, max_iter, mat.harden, mat.table.shape[0],
              mat.nlgeom, int(mat.plastic)], dtype=np.int32).tofile(f)
    np.array([mat.E, mat.nu, *mat.plconst, converg], dtype=np.float64).tofile(f)
    mat.table.tofile(f)
    I.astype(np.int32).tofile(f)
    R.astype(np.float64).tofile(f)
    np.ascontiguousarray(coord, dtype=np.float64).tofile(f)
    np.ascontiguousarray(conn, dtype=np.int32).tofile(f)
    np.ascontiguousarray(bc_node, dtype=np.int32).tofile(f)
    np.ascontiguousarray(bc_dof, dtype=np.int32).tofile(f)
    np.ascontiguousarray(bc_val, dtype=np.float64).tofile(f)
    np.ascontiguousarray(cload, dtype=np.float64).tofile(f)
    if len(mats) > 1:
        np.array([len(mats)], dtype=np.int32).tofile(f)
        for m2 in mats[1:]:
            np.array([m2.E, m2.nu, *m2.plconst], dtype=np.float64).tofile(f)
            np.array([m2.harden, m2.table.shape[0], m2.nlgeom, int(m2.plastic)], dtype=np.int32).tofile(f)
            m2.table.tofile(f)
        np.ascontiguousarray(elem_mat, dtype=np.int32).tofile(f)


def run_nl_elements(mat, coord, conn, unode, dunode, state, workdir=None):
    """Reference STF_C3D8Bbar (before the first update), Update_C3D8Bbar, STF_C3D8Bbar again, element
    by element.  state: dict of stress_bak strain_bak stress strain (ne,8,6), plstrain fstat (ne,8),
    istat (ne,8) int32.  Returns ke_before, qf, ke_after (row-major (ne,24,24)) and the new state."""
    exe = os.path.join(REFDIR, "ref_nl")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    ne = conn.shape[0]
    I, R = default_params()
    z = np.zeros(0)
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            _write_nl_head(f, 1, mat, coord, conn, z, z, z, np.zeros(coord.size), 1, 1, 1e-6, I, R)
            np.ascontiguousarray(unode, dtype=np.float64).tofile(f)
            np.ascontiguousarray(dunode, dtype=np.float64).tofile(f)
            for k in ("stress_bak", "strain_bak", "stress", "strain", "plstrain", "fstat"):
                np.ascontiguousarray(state[k], dtype=np.float64).tofile(f)
            np.ascontiguousarray(state["istat"], dtype=np.int32).tofile(f)
        p = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
        if p.returncode != 0 or not os.path.exists(fout):
            raise RuntimeError("ref_nl failed: " + p.stdout)
        with open(fout, "rb") as f:
            ke0 = np.fromfile(f, dtype=np.float64, count=576 * ne).reshape(ne, 24, 24).transpose(0, 2, 1).copy()
            qf = np.fromfile(f, dtype=np.float64, count=24 * ne).reshape(ne, 24)
            ke1 = np.fromfile(f, dtype=np.float64, count=576 * ne).reshape(ne, 24, 24).transpose(0, 2, 1).copy()
            out = dict(state)
            out["stress"] = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            out["strain"] = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            out["fstat"] = np.fromfile(f, dtype=np.float64, count=8 * ne).reshape(ne, 8)
            out["istat"] = np.fromfile(f, dtype=np.int32, count=8 * ne).reshape(ne, 8)
    return ke0, qf, ke1, out


def run_nl_steps(mat, coord, conn, bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R,
                 threads=2, workdir=None, timeout=None, elem_mat=None):
    """Reference load-step loop (see oracle/ref_nl_driver.f90).  Returns dict(log, unode, qforce, state)."""
    exe = os.path.join(REFDIR, "ref_nl")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    ne, nn = conn.shape[0], coord.shape[0]
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            _write_nl_head(f, 2, mat, coord, conn, bc_node, bc_dof, bc_val, cload, nsub, max_iter, converg, I, R, elem_mat)
        p = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)), timeout=timeout)
        if p.returncode != 0 or not os.path.exists(fout):
            raise RuntimeError("ref_nl failed: " + p.stdout[-2000:])
        with open(fout, "rb") as f:
            nlog = int(np.fromfile(f, dtype=np.int32, count=1)[0])
            log = np.fromfile(f, dtype=np.float64, count=7 * nlog).reshape(nlog, 7)
            out = {"log": log, "stdout": p.stdout}
            out["unode"] = np.fromfile(f, dtype=np.float64, count=3 * nn)
            out["qforce"] = np.fromfile(f, dtype=np.float64, count=3 * nn)
            st = {}
            st["stress"] = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            st["strain"] = np.fromfile(f, dtype=np.float64, count=48 * ne).reshape(ne, 8, 6)
            st["plstrain"] = np.fromfile(f, dtype=np.float64, count=8 * ne).reshape(ne, 8)
            st["fstat"] = np.fromfile(f, dtype=np.float64, count=8 * ne).reshape(ne, 8)
            st["istat"] = np.fromfile(f, dtype=np.int32, count=8 * ne).reshape(ne, 8)
            out["state"] = st
    return out
