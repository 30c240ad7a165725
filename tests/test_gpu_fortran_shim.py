"""The drop-in boundary end to end: a Fortran program holding the REFERENCE's own
hecmwST_matrix / hecmwST_local_mesh calls `hecmw_solve(hecMESH, hecMAT)` -- as
fistr1/src/lib/solve_LINEQ.f90:22 does -- and that name now resolves to
frontistr_amd/shim/hecmw_solver_hip.f90 -> libfistr_hip.so (oracle/_ref/shim_solve, linked by
oracle/build_ref.py from the reference's objects + our shim).  Output must match the golden
vectors the unmodified reference produced."""
import numpy as np
import pytest

from conftest import golden_matrix, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deck", ["cube4", "cube3s"])
@pytest.mark.parametrize("meth,pc,thr", [(1, 3, 1), (1, 1, 4)])
def test_fistr_side_call_through_shim(deck, meth, pc, thr):
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built (needs /root/reference at build time)")
    g = load_golden(deck)
    A = golden_matrix(g)
    I, R = refrun.default_params(method=meth, precond=pc)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve")
    assert r["returncode"] == 0, r["stdout"][-2000:]
    assert "reference CPU solver used" not in r["stdout"]   # the GPU path ran
    assert r["banner"] == "### 3x3 BLOCK CG, %s, 1" % {3: "DIAG", 1: "SSOR"}[pc]
    tag = "sol_m%d_p%d_t%d_" % (meth, pc, thr)
    x_ref, h_ref = g[tag + "X"], g[tag + "hist"]
    assert np.abs(r["X"] - x_ref).max() < 1e-8 * np.abs(x_ref).max()
    h = np.array([v for _, v in r["history"]])            # the shim prints the reference's ITERLOG format
    assert abs(len(h) - len(h_ref)) <= 1
    k = min(10, len(h), len(h_ref))
    assert np.all(np.abs(h[:k] - h_ref[:k]) <= 2e-6 * h_ref[:k])
    assert r["Iarray"][80] == 1 and r["Iarray"][96] == 0 and r["Iarray"][97] == 0
    assert abs(r["rel_resid"] - float(g[tag + "rel_resid"])) <= 0.5 * float(g[tag + "rel_resid"]) + 1e-12


def test_shim_cpu_escape_hatch():
    """HECMW_GPU=0 keeps the reference's CPU path inside the same binary."""
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    g = load_golden("cube4")
    A = golden_matrix(g)
    I, R = refrun.default_params(method=1, precond=3)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve", extra_env={"HECMW_GPU": "0"})
    assert "reference CPU solver used" in r["stdout"]        # announced, never silent
    assert np.array_equal(r["X"], g["sol_m1_p3_t1_X"])      # bit-for-bit the reference


def test_shim_refuses_what_is_not_on_the_gpu_path():
    """A preconditioner outside SSOR / DIAG / ILU(0) is refused (abort) -- no default CPU routing -- and runs on the reference's
    CPU solver only when HECMW_GPU_UNSUPPORTED=reference asks for that."""
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    A = golden_matrix(load_golden("cube4"))
    I, R = refrun.default_params(method=1, precond=11)      # block ILU(1)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve")
    assert "libfistr_hip-E: not on the GPU path" in r["stdout"] and "X" not in r     # hecmw_abort: no solution is written
    r = refrun.run_solve(A, I, R, exe_name="shim_solve", extra_env={"HECMW_GPU_UNSUPPORTED": "reference"})
    assert r["returncode"] == 0 and "reference CPU solver used" in r["stdout"] and r["Iarray"][80] == 1


@pytest.mark.parametrize("nd,meth,pc", [(1, 1, 3), (2, 2, 1), (6, 1, 1)])
def test_shim_generic_block_sizes(nd, meth, pc):
    """hecMAT%NDOF /= 3 through the same Fortran call: the generic-block GPU path, against the reference's golden vectors."""
    from nn_cases import nn_system, nn_tag
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    g = load_golden("nn")
    A = nn_system(nd)
    I, R = refrun.default_params(method=meth, precond=pc)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve")
    assert r["returncode"] == 0 and "reference CPU solver used" not in r["stdout"], r["stdout"][-2000:]
    assert r["banner"] == "### %dx%d BLOCK %s, %s, 1" % (nd, nd, {1: "CG", 2: "BiCGSTAB"}[meth], {3: "DIAG", 1: "SSOR"}[pc])
    tag = nn_tag(nd, meth, pc)
    assert abs(r["iter"] - int(g[tag + "iter"])) <= 1 if "iter" in r else True
    assert np.abs(r["X"] - g[tag + "X"]).max() <= 1e-8 * np.abs(g[tag + "X"]).max() and r["Iarray"][80] == 1


def test_shim_recycle_policy_sequence():
    """The Fortran program solves six times (values change, Iarray(97) = 1 raised each time): through the shim the GPU path applies
    the reference's recycle policy -- iteration counts of the unmodified reference (tests/golden/recycle.npz) +-1."""
    import test_oracle_golden as T
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    g = load_golden("recycle")
    A = golden_matrix(load_golden("cube4"))
    I, R = refrun.default_params(method=1, precond=1)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve", mode=4, nrepeat=6)
    tag = T.recycle_tag("cube4", 1, 1)
    assert r["returncode"] == 0 and "reference CPU solver used" not in r["stdout"]
    assert np.array_equal(r["Iarray"][95:98], g[tag + "Iarray"][95:98])
    assert np.abs(r["X"] - g[tag + "X"]).max() <= 1e-7 * np.abs(g[tag + "X"]).max()


@pytest.mark.parametrize("nd,meth,pc,scal", [(2, 3, 3, 0), (4, 4, 1, 0), (1, 1, 1, 1), (6, 2, 3, 1)])
def test_shim_generic_blocks_other_methods_and_scaling(nd, meth, pc, scal):
    """GMRES / GPBiCG and SCALING=YES for hecMAT%NDOF /= 3 through the Fortran call, against the CPU oracle (bit-identical to the
    reference on these systems, tests/test_oracle_nn.py)."""
    from nn_cases import nn_system
    from oracle import pyoracle, refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    A = nn_system(nd)
    I, R = refrun.default_params(method=meth, precond=pc)
    I[6] = scal
    o = pyoracle.solve_iterative(A, I, R, nthreads=4)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve")
    assert r["returncode"] == 0 and "reference CPU solver used" not in r["stdout"] and "libfistr_hip-E" not in r["stdout"], r["stdout"][-1500:]
    assert r["banner"] == "### %dx%d BLOCK %s, %s, 1" % (nd, nd, {1: "CG", 2: "BiCGSTAB", 3: "GMRES", 4: "GPBiCG"}[meth], {3: "DIAG", 1: "SSOR"}[pc])
    assert r["Iarray"][80] == 1 and np.abs(r["X"] - o["X"]).max() <= 1e-7 * np.abs(o["X"]).max()


def _summary_labels(stdout):
    """Labels of the TIMELOG block (hecmw_solver_Iterative.f90:192-208), in order."""
    out, on = [], False
    for ln in stdout.splitlines():
        if ln.startswith("### summary of linear solver"):
            on = True
            out.append(ln.strip())
            continue
        if on:
            if "iterations" in ln and ":" not in ln:
                out.append("iterations")
            elif ":" in ln and ln.startswith("    "):
                out.append(ln.split(":")[0].strip())
                if "work ratio" in ln:
                    break
    return out


@pytest.mark.parametrize("meth,pc", [(1, 1), (2, 10)])
def test_shim_stdout_is_the_references(meth, pc):
    """SURVEY 8(b) 'stdout lines to reproduce': the same deck through the unmodified reference (oracle/_ref/ref_solve_omp) and
    through the shim: identical banner ('### 3x3 BLOCK CG, SSOR, 1', hecmw_solver_Iterative.f90:418-419), the same ITERLOG
    lines ('(i7,1pe16.6)', 6 printed digits equal on the first ten), the '### Relative residual =' line and the TIMELOG
    summary block with the reference's labels in the reference's order (the times themselves differ, the iteration line does
    not)."""
    from oracle import refrun
    if not (refrun.have_ref("shim_solve") and refrun.have_ref("ref_solve_omp")):
        pytest.skip("oracle/_ref not built")
    A = golden_matrix(load_golden("cube4"))
    I, R = refrun.default_params(method=meth, precond=pc, iterlog=1, timelog=1)
    ref = refrun.run_solve(A, I, R, threads=4)
    shm = refrun.run_solve(A, I, R, exe_name="shim_solve")
    assert shm["returncode"] == 0, shm["stdout"][-2000:]
    assert ref["banner"] == shm["banner"] == "### 3x3 BLOCK %s, %s, 1" % ({1: "CG", 2: "BiCGSTAB"}[meth], {1: "SSOR", 10: "ILU(0)"}[pc])
    rl = [ln for ln in ref["stdout"].splitlines() if refrun.HIST_RE.match(ln)]
    sl = [ln for ln in shm["stdout"].splitlines() if refrun.HIST_RE.match(ln)]
    assert abs(len(rl) - len(sl)) <= (1 if meth == 1 else 2)
    for a, b in list(zip(rl, sl))[:10]:
        assert a[:7] == b[:7] and len(a) == len(b)                                   # same iteration column, same width
        assert abs(float(a[7:]) - float(b[7:])) <= 2e-6 * float(a[7:])
    rr = [ln for ln in ref["stdout"].splitlines() if ln.startswith("### Relative residual =")]
    sr = [ln for ln in shm["stdout"].splitlines() if ln.startswith("### Relative residual =")]
    assert len(rr) == len(sr) == 1 and len(rr[0]) == len(sr[0])
    labels = _summary_labels(ref["stdout"])
    assert labels == _summary_labels(shm["stdout"]) and labels[0] == "### summary of linear solver" and "solver/matvec" in labels
    assert shm["t_matvec"] > 0 and shm["t_precond"] > 0 and shm["t_solver"] >= shm["t_matvec"]     # measured, not zeros
    assert abs(shm["iter"] - ref["iter"]) <= (1 if meth == 1 else 2)
    # order of the channels: banner, history, residual line, summary
    pos = [shm["stdout"].index(s) for s in (shm["banner"], sl[0], sr[0], "### summary of linear solver")]
    assert pos == sorted(pos)


def test_resident_matvec_binding_notices_a_solve_in_between():
    """HECMW_GPU_MATVEC=resident re-uses the values the hecmw_matvec binding uploaded -- but the binding shares the rank's one
    device context with hecmw_solve, and a solve in between puts ITS matrix there.  hecmw_matvec, a hecmw_solve of
    another matrix of the same shape (D scaled by 2), hecmw_matvec with the first matrix again: the second product must be the
    first matrix's (= the reference CPU product), not the solver's."""
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    A = golden_matrix(load_golden("cube3s"))
    A.X = np.sin(0.37 * np.arange(3 * A.NP) + 0.1)
    I, R = refrun.default_params(method=1, precond=3, iterlog=0, timelog=0)
    cpu = refrun.run_solve(A, I, R, mode=2, exe_name="shim_solve")                       # the reference's own product
    res = refrun.run_solve(A, I, R, mode=5, exe_name="shim_solve", extra_env={"HECMW_GPU_MATVEC": "resident"})
    assert cpu["returncode"] == 0 and res["returncode"] == 0, res["stdout"][-2000:]
    n3 = 3 * A.N
    assert np.abs(res["X"][:n3] - cpu["X"][:n3]).max() < 1e-13 * np.abs(cpu["X"][:n3]).max()


@pytest.mark.parametrize("k", [0, 5])
def test_shim_stdout_of_the_retry_loop_is_the_references(k):
    """The auto-SIGMA_DIAG / METHOD2 loop (hecmw_solver_Iterative.f90:117-157) prints its banner before EVERY pass (:125), the
    pass's ITERLOG lines, and ' Increasing SIGMA_DIAG to <value>' before a SIGMA_DIAG retry (:149).  The same diverging deck
    (tests/golden/retry.npz case k: CG + ILU(0), one diagonal block scaled, SIGMA_DIAG = -1 'auto'; k = 5 with METHOD2 = 2)
    through the unmodified reference and through the shim: the same sequence of banners and retry lines, character for
    character (the shim replays them from fx_solve_attempts), the same pass lengths."""
    from oracle import refrun
    if not (refrun.have_ref("shim_solve") and refrun.have_ref("ref_solve")):
        pytest.skip("oracle/_ref not built")
    g = load_golden("retry")
    blk, scale, sigma, m2 = g["c%d_case" % k]
    A = golden_matrix(load_golden("cube4"))
    A.D = A.D.copy()
    A.D[9 * int(blk):9 * int(blk) + 9] *= scale
    I, R = refrun.default_params(method=1, precond=10, maxit=500, iterlog=1, timelog=0)
    R[1] = sigma
    I[7] = int(m2)
    ref = refrun.run_solve(A, I.copy(), R.copy(), threads=1)
    shm = refrun.run_solve(A, I.copy(), R.copy(), exe_name="shim_solve")

    def channel(out):
        return [ln for ln in out.splitlines() if ln.startswith("### 3x3 BLOCK") or "Increasing SIGMA_DIAG" in ln]

    def passes(out):
        lens, cur = [], None
        for ln in out.splitlines():
            if ln.startswith("### 3x3 BLOCK"):
                if cur is not None:
                    lens.append(cur)
                cur = 0
            elif cur is not None and refrun.HIST_RE.match(ln):
                cur += 1
        return lens + [cur]

    rc, sc = channel(ref["stdout"]), channel(shm["stdout"])
    assert int(g["c%d_n_sigma_msgs" % k]) == sum("Increasing" in ln for ln in rc) >= 1       # the deck does retry
    assert rc == sc                                          # banners and 'Increasing SIGMA_DIAG to' lines, character for character
    rp, sp = passes(ref["stdout"]), passes(shm["stdout"])
    assert list(g["c%d_attempts" % k]) == rp and len(sp) == len(rp)
    for a, b in zip(rp[:-1], sp[:-1]):                       # the diverging passes stop at the same line (+-1: the sign test on rounding-level rho)
        assert abs(a - b) <= 1
    assert shm["Iarray"][80] == ref["Iarray"][80] and shm["Iarray"][81] == ref["Iarray"][81]


def _self_neighbour_system(seed=5):
    """One subdomain of a 2x1x1 cube split whose only neighbour is the rank itself (periodic tables: halo node k receives
    internal node export_item[k]), random diagonally dominant 3x3 blocks."""
    from frontistr_amd import hecmw as hip
    from frontistr_amd.partition import cube_subdomain
    sub = cube_subdomain(6, (2, 1, 1), 0)
    hm = sub.hecmesh(hip)
    hm.elem_node_item = sub.conn.ravel()
    m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
    N, NP = m.N, m.NP
    rng = np.random.default_rng(seed)
    m.D = rng.standard_normal(9 * NP); m.AL = 0.1 * rng.standard_normal(9 * m.NPL); m.AU = 0.1 * rng.standard_normal(9 * m.NPU)
    m.D.reshape(NP, 3, 3)[:] += 6.0 * np.eye(3)
    m.B = rng.standard_normal(3 * NP); m.X = np.zeros(3 * NP)
    n_imp = NP - N
    exp_item = (1 + rng.permutation(N)[:n_imp]).astype(np.int32)
    comm = dict(PETOT=1, my_rank=0, neighbor_pe=np.zeros(1, dtype=np.int32), import_index=np.array([0, n_imp], dtype=np.int32),
                export_index=np.array([0, n_imp], dtype=np.int32), import_item=hm.import_item.copy(), export_item=exp_item)
    return m, comm


def _dense(m, comm, periodic):
    """Dense 3N x 3N operator of the internal rows; halo columns folded onto their exporting node (periodic) or dropped."""
    N, NP = m.N, m.NP
    src = np.arange(NP)
    src[comm["import_item"] - 1] = comm["export_item"] - 1
    K = np.zeros((3 * N, 3 * N))
    for i in range(N):
        K[3 * i:3 * i + 3, 3 * i:3 * i + 3] += m.D[9 * i:9 * i + 9].reshape(3, 3)
        for idx, item, val in ((m.indexL, m.itemL, m.AL), (m.indexU, m.itemU, m.AU)):
            for j in range(idx[i], idx[i + 1]):
                c = item[j] - 1
                if c >= N:
                    if not periodic:
                        continue
                    c = src[c]
                K[3 * i:3 * i + 3, 3 * c:3 * c + 3] += val[9 * j:9 * j + 9].reshape(3, 3)
    return K


@pytest.mark.parametrize("transport", ["rccl", "mpi"])
def test_shim_bootstraps_the_transport_of_a_decomposed_run(transport):
    """A rank with neighbour tables through the Fortran driver: the shim establishes the transport on the first call.
    HECMW_GPU_TRANSPORT=rccl: fx_comm_unique_id -> hecmw_bcast_C -> fx_comm_init, the grouped ncclSend / ncclRecv exchange runs
    against the rank itself (periodic tables) and the solution is that of the periodic operator.  HECMW_GPU_TRANSPORT=mpi:
    fx_comm_set_host_callbacks wired to hecmw_update_m_R / hecmw_allreduce_R of the reference; in the HECMW_SERIAL build linked
    here those are no-ops (hecmw_comm_f.F90 `#ifndef HECMW_SERIAL`), so the halo part stays zero and the solution is that of
    the operator without halo columns -- which proves the callbacks, not a silent skip, served the exchange."""
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    m, comm = _self_neighbour_system()
    N = m.N
    A = refrun.BSR(m.N, m.NP, m.indexL, m.itemL, m.indexU, m.itemU, m.D, m.AL, m.AU, m.B)
    I, R = refrun.default_params(method=2, precond=3, maxit=200, tol=1e-10)
    r = refrun.run_solve(A, I, R, exe_name="shim_solve", comm=comm, extra_env={"HECMW_GPU_TRANSPORT": transport})
    assert r["returncode"] == 0, r["stdout"][-3000:]
    assert ("RCCL communicator over 1 rank" if transport == "rccl" else "through hecmw_update_m_R / hecmw_allreduce_R") in r["stdout"]
    K = _dense(m, comm, periodic=(transport == "rccl"))
    x = np.linalg.solve(K, m.B[:3 * N])
    assert r["Iarray"][80] == 1
    assert np.abs(r["X"][:3 * N] - x).max() < 1e-7 * np.abs(x).max()
    if transport == "rccl":      # hecmw_update_m_R(X) at the end: the halo part holds the exporting nodes' values
        xh = r["X"].reshape(-1, 3)
        assert np.abs(xh[comm["import_item"] - 1] - xh[comm["export_item"] - 1]).max() == 0.0


def test_hecmw_matvec_binding_through_the_reference_module():
    """hecmw_matvec (module hecmw_solver_las, las/hecmw_solver_las.f90:57-77) with the four-line patch of INTEGRATION.md
    section 2 and frontistr_amd/shim/hecmw_matvec_hip.f90: the Fortran driver's `call hecmw_matvec(hecMESH, hecMAT, X, Y, t)`
    lands in fx_matvec (HECMW_GPU_MATVEC=1) and equals the reference's own CPU product (same binary, variable unset);
    on a subdomain with halo tables the halo part of X is updated as hecmw_update_3_R does."""
    from oracle import refrun
    if not refrun.have_ref("shim_solve"):
        pytest.skip("oracle/_ref/shim_solve not built")
    g = load_golden("cube3s")
    A = golden_matrix(g)
    A.X = np.sin(0.37 * np.arange(3 * A.NP) + 0.1)
    I, R = refrun.default_params()
    cpu = refrun.run_solve(A, I, R, mode=2, exe_name="shim_solve")
    gpu = refrun.run_solve(A, I, R, mode=2, exe_name="shim_solve", extra_env={"HECMW_GPU_MATVEC": "1"}, nrepeat=2)
    res = refrun.run_solve(A, I, R, mode=2, exe_name="shim_solve", extra_env={"HECMW_GPU_MATVEC": "resident"}, nrepeat=3)
    assert cpu["returncode"] == 0 and gpu["returncode"] == 0 and res["returncode"] == 0, gpu["stdout"][-2000:]
    n3 = 3 * A.N
    assert np.abs(gpu["X"][:n3] - cpu["X"][:n3]).max() < 1e-13 * np.abs(cpu["X"][:n3]).max()
    assert np.array_equal(res["X"][:n3], gpu["X"][:n3])
    # a subdomain: periodic self-neighbour tables over RCCL
    m, comm = _self_neighbour_system(seed=9)
    A2 = refrun.BSR(m.N, m.NP, m.indexL, m.itemL, m.indexU, m.itemU, m.D, m.AL, m.AU, m.B)
    A2.X = np.cos(0.21 * np.arange(3 * m.NP))
    r = refrun.run_solve(A2, I, R, mode=2, exe_name="shim_solve", comm=comm, extra_env={"HECMW_GPU_MATVEC": "1"})
    assert r["returncode"] == 0, r["stdout"][-2000:]
    K = _dense(m, comm, periodic=True)
    y = K @ A2.X[:3 * m.N]
    assert np.abs(r["X"][:3 * m.N] - y).max() < 1e-12 * np.abs(y).max()
    xh = np.concatenate([A2.X[:3 * m.N], r["X"][3 * m.N:]]).reshape(-1, 3)      # the driver returns X's halo part in the tail
    assert np.array_equal(xh[comm["import_item"] - 1], xh[comm["export_item"] - 1])
