"""Oracle (CPU restatement, oracle/fstr_nl_oracle.c) of the nonlinear C3D8 B-bar path against the
reference routines themselves (oracle/_ref/ref_nl, built from /root/reference by oracle/build_ref.py)
where that binary exists, and against the committed fixtures it produced (tests/golden/nl_*.npz)
everywhere.  SURVEY §8(f)-2."""
import os

import numpy as np
import pytest

from frontistr_amd.mesh import CubeMesh
from oracle import pyoracle, refrun

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TABLE = [[450, 0], [608, 0.05], [679, 0.1], [732, 0.2], [752, 0.3], [766, 0.4], [780, 0.5]]   # tutorial/05_plastic_cylinder


def materials():
    return {
        "elastic_ul": refrun.Material(206900.0, 0.29, plastic=False, nlgeom=2),
        "elastic_tl": refrun.Material(206900.0, 0.29, plastic=False, nlgeom=1),
        "mises_multilinear_ul": refrun.Material(206900.0, 0.29, plastic=True, harden=1, table=TABLE, nlgeom=2),
        "mises_bilinear_ul": refrun.Material(1.0e5, 0.3, plastic=True, harden=0, plconst=(1000.0, 2.0e3, 0.0), nlgeom=2),
        "mises_swift_tl": refrun.Material(1.0e5, 0.3, plastic=True, harden=2, plconst=(0.01, 2000.0, 0.2), nlgeom=1),
        "mises_ramberg_inf": refrun.Material(1.0e5, 0.3, plastic=True, harden=3, plconst=(0.005, 800.0, 5.0), nlgeom=0),
    }


def element_case(name, seed=0):
    """A 2x2x2 skewed block with random displacement, increment and history (some points plastic)."""
    mat = materials()[name]
    m = CubeMesh(2, skew=0.12)
    rng = np.random.default_rng(seed)
    unode = 0.02 * rng.standard_normal(m.ndof)
    dunode = 0.01 * rng.standard_normal(m.ndof)
    st = pyoracle.new_state(m.n_elem)
    scale = 300.0 if mat.plastic else 100.0
    st["stress_bak"] = scale * rng.standard_normal((m.n_elem, 8, 6))
    st["strain_bak"] = 1e-3 * rng.standard_normal((m.n_elem, 8, 6))
    st["stress"] = st["stress_bak"] + 10 * rng.standard_normal((m.n_elem, 8, 6))
    st["strain"] = st["strain_bak"].copy()
    if mat.plastic:
        st["plstrain"] = np.abs(0.02 * rng.standard_normal((m.n_elem, 8)))
        st["fstat"] = st["plstrain"] + np.abs(0.005 * rng.standard_normal((m.n_elem, 8)))
        st["istat"] = (rng.random((m.n_elem, 8)) < 0.5).astype(np.int32)
    return mat, m, unode, dunode, st


def check_elements(got, want):
    ke0, qf, ke1, st = got
    rke0, rqf, rke1, rst = want
    for a, b, tag in ((ke0, rke0, "ke before"), (qf, rqf, "qf"), (ke1, rke1, "ke after"),
                      (st["stress"], rst["stress"], "stress"), (st["strain"], rst["strain"], "strain"),
                      (st["fstat"], rst["fstat"], "fstat")):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11 * np.abs(b).max(), err_msg=tag)
    assert np.array_equal(st["istat"], rst["istat"])


@pytest.mark.parametrize("name", list(materials()))
def test_elements_vs_golden(name):
    mat, m, unode, dunode, st = element_case(name)
    g = np.load(os.path.join(GOLD, "nl_elements_%s.npz" % name))
    want = (g["ke0"], g["qf"], g["ke1"], {k: g[k] for k in ("stress", "strain", "fstat", "istat")})
    check_elements(pyoracle.nl_elements(mat, m.coord, m.conn, unode, dunode, st), want)
    # the reference's latched MatlMatrix flag: after the first update the tangent of a plastic point is elastic
    if mat.plastic and st["istat"].any():
        assert np.abs(g["ke0"] - g["ke1"]).max() > 1e-3 * np.abs(g["ke0"]).max()


@pytest.mark.skipif(not refrun.have_ref("ref_nl"), reason="oracle/_ref/ref_nl not built (no /root/reference)")
@pytest.mark.parametrize("name", list(materials()))
def test_elements_vs_reference(name):
    mat, m, unode, dunode, st = element_case(name, seed=3)
    want = refrun.run_nl_elements(mat, m.coord, m.conn, unode, dunode, st)
    check_elements(pyoracle.nl_elements(mat, m.coord, m.conn, unode, dunode, st), want)


def step_case(name):
    mat = materials()[name]
    m = CubeMesh(3, skew=0.1)
    bn, bd, bv = m.dirichlet()
    tn = np.repeat(m.top_nodes, 3).astype(np.int32)
    td = np.tile(np.array([1, 2, 3], dtype=np.int32), m.top_nodes.size)
    tv = np.tile(np.array([0.06, 0.0, 0.12]), m.top_nodes.size)
    bc = (np.concatenate([bn, tn]), np.concatenate([bd, td]), np.concatenate([bv, tv]))
    cload = np.zeros(m.ndof)
    cload[3 * (m.conn[13, 6] - 1) + 1] = 50.0          # a nodal load on an interior node as well
    I, R = refrun.default_params(method=1, precond=3, tol=1e-10, iterlog=0, timelog=0)
    return mat, m, bc, cload, I, R


STEP_CASES = ["mises_multilinear_ul", "mises_bilinear_ul", "elastic_tl"]
# Newton tolerance per case: the reference's latched elastic tangent converges linearly, so the
# plastic decks either stop on a loose CONVERG (multilinear) or run into max_iter (bilinear).
STEP_CONVERG = {"mises_multilinear_ul": 5e-2, "mises_bilinear_ul": 1e-6, "elastic_tl": 1e-6}


def check_steps(model, log, want):
    wl = want["log"]
    assert log.shape[0] == wl.shape[0]
    assert np.array_equal(log[:, :2], wl[:, :2])                      # same Newton iteration counts per substep
    np.testing.assert_allclose(log[:, 3:], wl[:, 3:], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(model.unode, want["unode"], rtol=0, atol=1e-9 * np.abs(want["unode"]).max())
    s = want["stress"]
    np.testing.assert_allclose(model.state["stress"], s, rtol=0, atol=1e-8 * np.abs(s).max())
    np.testing.assert_allclose(model.state["plstrain"], want["plstrain"], rtol=0, atol=1e-10)
    assert np.array_equal(model.state["istat"], want["istat"])


SECTION_CASES = ["ul", "mixed"]


def sections_case(name):
    """The bilinear step deck with several sections: (materials, elem_mat, mesh, bc, cload, I, R, converg)."""
    mats = materials()
    mat, m, bc, cload, I, R = step_case("mises_bilinear_ul")
    ne = m.conn.shape[0]
    if name == "ul":
        ms = [mat, mats["mises_multilinear_ul"], mats["elastic_ul"]]
    else:
        ms = [mat, mats["elastic_tl"]]
    emat = (1 + (np.arange(ne) % len(ms))).astype(np.int32)
    return ms, emat, m, bc, cload, I, R, STEP_CONVERG["mises_bilinear_ul"]


@pytest.mark.parametrize("name", SECTION_CASES)
def test_load_steps_several_sections_vs_golden(name):
    """Three / two materials in one element group, one of the decks with mixed NLGEOM flags
    (tests/golden/make_nl_sections_golden.py ran them through the reference routines)."""
    ms, emat, m, bc, cload, I, R, conv = sections_case(name)
    g = np.load(os.path.join(GOLD, "nl_steps_sections_%s.npz" % name))
    assert np.array_equal(g["elem_mat"], emat)
    model = pyoracle.NonlinearModel(ms, m.coord, m.conn, elem_mat=emat)
    log = model.run_steps(*bc, cload, 3, 12, conv, I, R, nthreads=2)
    check_steps(model, log, g)
    assert g["plstrain"].max() > 1e-3 and (g["plstrain"].reshape(-1, 8)[emat == len(ms)] == 0).all()   # the elastic section stays elastic


@pytest.mark.parametrize("name", STEP_CASES)
def test_load_steps_vs_golden(name):
    mat, m, bc, cload, I, R = step_case(name)
    g = np.load(os.path.join(GOLD, "nl_steps_%s.npz" % name))
    model = pyoracle.NonlinearModel(mat, m.coord, m.conn)
    log = model.run_steps(*bc, cload, 3, 12, STEP_CONVERG[name], I, R, nthreads=2)
    check_steps(model, log, g)
    if mat.plastic:
        assert g["plstrain"].max() > 1e-3


def necking_case():
    """The reference's own deck tutorial/05_plastic_cylinder (mesh + control data in the fixture)."""
    g = np.load(os.path.join(GOLD, "nl_necking.npz"))
    mat = refrun.Material(206900.0, 0.29, plastic=True, harden=1, table=g["table"], nlgeom=2)
    nsub, frac = int(g["nsub"]), float(g["nsub"]) / float(g["nsub_total"])
    bc = (g["bc_node"], g["bc_dof"], g["bc_val"] * frac)
    I, R = refrun.default_params(method=1, precond=1, maxit=2000, tol=1e-8, iterlog=0, timelog=0)
    return g, mat, bc, nsub, int(g["max_iter"]), float(g["converg"]), I, R


def test_plastic_cylinder_tutorial_vs_golden():
    g, mat, bc, nsub, max_iter, converg, I, R = necking_case()
    model = pyoracle.NonlinearModel(mat, g["coord"], g["conn"])
    log = model.run_steps(*bc, np.zeros(3 * g["coord"].shape[0]), nsub, max_iter, converg, I, R, nthreads=2)
    check_steps(model, log, g)
    assert log.shape[0] == 46 and int(g["istat"].sum()) == g["istat"].size      # 36 + 5 + 5 Newton iterations, fully plastic


def exI_case():
    """examples/static/exI: exA's A361 mesh under I300.cnt (NLGEOM, elastic, 10 substeps, CG + DIAG 1e-8)."""
    from conftest import load_golden
    d, e = load_golden("exA_A361"), np.load(os.path.join(GOLD, "exI_A361_expect.npz"))
    mat = refrun.Material(float(e["E"]), float(e["nu"]), plastic=False, nlgeom=1)     # elastic under NLGEOM: TOTALLAG
    bc = (d["bc_node"], d["bc_dof"], d["bc_val"])
    I, R = refrun.default_params(method=1, precond=3, maxit=10000, tol=1e-8, iterlog=0, timelog=0)
    return d, e, mat, bc, I, R


def check_exI_extrema(unodes, e):
    for s, u in enumerate(unodes):
        U = u.reshape(-1, 3)
        for c in range(3):
            mx, mn = e["extrema"][s, c]
            assert abs(U[:, c].max() - mx) <= 1e-4 and abs(U[:, c].min() - mn) <= 1e-4, (s + 1, c)   # test_FrontISTR.rb:10


def test_exI_known_answer():
    """The reference's own known answer for the geometrically nonlinear path: displacement extrema of all 10 steps of
    exI/A361_correct.log at the reference harness' tolerance."""
    d, e, mat, bc, I, R = exI_case()
    model = pyoracle.NonlinearModel(mat, d["coord"], d["conn"])
    nsub = int(e["substeps"])
    unodes = []
    for sub in range(1, nsub + 1):          # run_steps one substep at a time to look at every step
        model.run_steps(bc[0], bc[1], bc[2], d["load"], 1, int(e["max_iter"]), float(e["converg"]), I, R, nthreads=1,
                        factors=((sub - 1) / nsub, sub / nsub))
        unodes.append(model.unode.copy())
    check_exI_extrema(unodes, e)


ONE_ELEM = ["mises", "swift", "ramberg"]


def one_elem_case(name):
    """examples/static/1elem/{mises,swift,ramberg}: unit cube, uniaxial stretch in 10 substeps."""
    g = np.load(os.path.join(GOLD, "nl_1elem.npz"))
    mat = refrun.Material(float(g[name + "_E"]), float(g[name + "_nu"]), plastic=True, harden=int(g[name + "_harden"]),
                          plconst=tuple(g[name + "_plconst"]), nlgeom=2)
    bc = (g[name + "_bc_node"], g[name + "_bc_dof"], g[name + "_bc_val"])
    I, R = refrun.default_params(method=1, precond=1, maxit=10000, tol=1e-12, iterlog=0, timelog=0)
    want = {k: g[name + "_" + k] for k in ("log", "unode", "stress", "plstrain", "istat")}
    return g, mat, bc, float(g[name + "_converg"]), I, R, want


def uniaxial_yield_stress(name, mat, pl):
    """The hardening law itself (calCurrYield, Elastoplastic.f90:254-292) at the final plastic strain: under the decks'
    uniaxial stress state sigma_xx must sit on it -- a check that does not involve the reference's output."""
    if name == "mises":
        return mat.plconst[0] + mat.plconst[1] * pl
    if name == "swift":
        return mat.plconst[1] * (mat.plconst[0] + pl) ** mat.plconst[2]
    return mat.plconst[1] * (pl / mat.plconst[0]) ** (1.0 / mat.plconst[2])


@pytest.mark.parametrize("name", ONE_ELEM)
def test_one_element_plasticity_decks(name):
    g, mat, bc, converg, I, R, want = one_elem_case(name)
    model = pyoracle.NonlinearModel(mat, g["coord"], g["conn"])
    log = model.run_steps(*bc, np.zeros(24), 10, 50, converg, I, R, nthreads=2)
    check_steps(model, log, want)
    s, pl = model.state["stress"][0], model.state["plstrain"][0]
    sy = uniaxial_yield_stress(name, mat, pl[0])
    assert np.abs(s[:, 0] - sy).max() < 2e-3 * sy and np.abs(s[:, 1:]).max() < 2e-3 * sy      # BackwardEuler's tol = 1e-3 on f
