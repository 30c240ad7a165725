"""The harness that judges fistr1 runs (oracle/fistr1_run.py = examples/test_FrontISTR.rb restated) and the reference's own
main program built here (oracle/_ref/fistr1_ref): it must reproduce the *_correct.log files the reference ships, otherwise
nothing measured with it means anything.  CPU only."""
import os

import pytest

from oracle import fistr1_run as f1


def _need(binary):
    if not f1.have(binary):
        if os.path.isdir("/root/reference"):
            pytest.fail("oracle/_ref/%s missing although /root/reference is here: run python oracle/build_ref.py --only fistr1" % binary)
        pytest.skip("oracle/_ref/%s not built (needs /root/reference at build time)" % binary)


def test_log_reader_reads_both_formats():
    old = f1.read_log(os.path.join(f1.DECKS, "exA", "A361_correct.log"))
    assert len(old) == 1 and old[0]["Node"]["U3"] == (0.0, -9.8430E-01) and old[0]["Node"]["E31"] == (-7.5570E-05, -1.2244E-03)
    assert old[0]["Element"]["SMS"] == (2.6641E+01, 2.4615E+00)
    new = f1.read_log(os.path.join(f1.DECKS, "t05", "necking_fistr1_ref_0.log"))
    assert len(new) == 9 and set(new[1]["Node"]) >= {"U1", "U2", "U3", "S33", "SMS"}
    assert f1.compare_step(old[0], old[0]) == []
    worse = {"Node": dict(old[0]["Node"], U3=(0.0, -9.8441E-01)), "Element": old[0]["Element"]}
    assert f1.compare_step(worse, old[0]) == [("Node", "U3", "min", -9.8441E-01, -9.8430E-01)]
    sta = f1.read_sta(os.path.join(f1.DECKS, "t05", "necking_fistr1_ref_FSTR.sta"))
    assert [r[3] for r in sta] == [36, 5, 5, 5, 5, 5, 5, 5, 50] and sta[8][2] == "1F" and "MAXITER" in sta[8][4]


def test_solver_card_rewrite(tmp_path):
    f1.prepare("exA", str(tmp_path), "A361.msh", "A300.cnt", method="BiCGSTAB", precond=10)
    s = open(tmp_path / "A300.cnt").read()
    assert "!SOLVER,METHOD=BiCGSTAB,PRECOND=10,ITERLOG=YES,TIMELOG=YES" in s and "VISUAL" not in s
    assert "!RESTART, FREQUENCY=100000" in s and s.rstrip().endswith("!END")


@pytest.mark.parametrize("deck,cnt,threads", [("exA", "A300.cnt", 1), ("exA", "A300.cnt", 4), ("exI", "I300.cnt", 4)])
def test_reference_program_reproduces_its_correct_logs(deck, cnt, threads):
    _need("fistr1_ref")
    r = f1.run_deck("fistr1_ref", deck, "A361.msh", cnt, threads=threads)
    assert r["returncode"] == 0 and "FrontISTR Completed !!" in r["stdout"], r["stdout"][-2000:]
    correct = f1.read_log(os.path.join(f1.DECKS, deck, "A361_correct.log"))
    got = r["log"][1:]                                  # block 0 is the all-zero summary of step 0
    assert len(got) == len(correct)
    for a, c in zip(got, correct):
        assert f1.compare_step(a, c) == []


def test_reference_program_on_the_plastic_cylinder():
    """tutorial/05 (configs[4]'s deck): 36 + 7 x 5 Newton iterations, stop at sub-step 9 -- SURVEY section 0."""
    _need("fistr1_ref")
    r = f1.run_deck("fistr1_ref", "t05", "necking.msh", "necking.cnt", threads=4)
    assert [x[3] for x in r["sta"][:8]] == [36, 5, 5, 5, 5, 5, 5, 5] and r["sta"][8][2] == "1F"
    want = f1.read_log(os.path.join(f1.DECKS, "t05", "necking_fistr1_ref_0.log"))
    assert len(r["log"]) == len(want) == 9
    for a, c in zip(r["log"], want):
        assert f1.compare_step(a, c) == []


def _static_models():
    import json
    with open(os.path.join(f1.DECKS, "static", "manifest.json")) as fh:
        return [tuple(x) for x in json.load(fh)]


@pytest.mark.parametrize("sub,model,mesh,cnt,ndof", _static_models(), ids=lambda v: str(v))
def test_reference_program_on_its_static_regression_decks(sub, model, mesh, cnt, ndof):
    """examples/static/exA ... exG + FbarElement (83 models: 2-D, solid, shell; linear, NLGEOM, F-bar): the unmodified program
    against the shipped *_correct.log, the baseline tests/test_gpu_fistr1.py holds fistr1_hip to."""
    _need("fistr1_ref")
    r = f1.run_deck("fistr1_ref", os.path.join("static", sub), mesh, cnt)
    assert r["returncode"] == 0 and "FrontISTR Completed !!" in r["stdout"], r["stdout"][-2000:]
    correct = f1.read_log(os.path.join(f1.DECKS, "static", sub, model + "_correct.log"))
    assert correct and r["log"] and f1.compare_step(r["log"][-1], correct[-1]) == []


def _heat_models():
    import json
    with open(os.path.join(f1.DECKS, "heat", "manifest.json")) as fh:
        return [tuple(x) for x in json.load(fh)]


@pytest.mark.parametrize("sub,model,mesh,cnt,ndof", _heat_models(), ids=lambda v: str(v))
def test_reference_program_on_its_heat_regression_decks(sub, model, mesh, cnt, ndof):
    """examples/heat/exM ... exT (80 models, NDOF = 1, CG + SSOR): the unmodified program against the shipped *_correct.log."""
    _need("fistr1_ref")
    r = f1.run_deck("fistr1_ref", os.path.join("heat", sub), mesh, cnt)
    assert r["returncode"] == 0, r["stdout"][-2000:]
    assert f1.heat_matches(r["heat"], f1.read_heat_log(os.path.join(f1.DECKS, "heat", sub, model + "_correct.log")))


TUTORIALS = [("t03", "cylinder.msh", "cylinder.cnt", [3, 3, 3, 3, 3]),
             ("t07", "cylinder.msh", "cylinder.cnt", [3] * 10),
             ("t08", "cylinder.msh", "cylinder.cnt", [3, 3, 3, 3, 2])]


@pytest.mark.parametrize("deck,mesh,cnt,newton", TUTORIALS, ids=[t[0] for t in TUTORIALS])
def test_reference_program_on_the_nonlinear_tutorials(deck, mesh, cnt, newton):
    """tutorial/03 (Mooney-Rivlin), 07 (viscoelastic), 08 (Norton creep): the stored expected output (4 threads: multicolour SSOR,
    tests/golden/make_tutorial_golden.py) is reproduced by the 1-thread run (natural-order SSOR) within the harness's 1e-4 --
    the preconditioner's ordering does not show at this tolerance, which is what lets the GPU runs be held to these files."""
    _need("fistr1_ref")
    r = f1.run_deck("fistr1_ref", deck, mesh, cnt, threads=1)
    assert [x[3] for x in r["sta"]] == newton
    want = f1.read_log(os.path.join(f1.DECKS, deck, cnt[:-4] + "_fistr1_ref_0.log"))
    assert len(r["log"]) == len(want) == len(newton) + 1
    for a, c in zip(r["log"], want):
        assert f1.compare_step(a, c) == []
