"""TEST INFRASTRUCTURE ONLY.  Run the reference's OWN main program on one of its decks and read what it wrote.

  oracle/_ref/fistr1_ref   fistr1 built from /root/reference unchanged (oracle/build_ref.py --only fistr1)
  oracle/_ref/fistr1_hip   the same program with module hecmw_solver (+ the hecmw_matvec patch) from frontistr_amd/shim/
                           -> libfistr_hip.so: fistr_main.f90:38-114 -> fstr_solve_NLGEOM -> fstr_Newton ->
                           solve_LINEQ (fistr1/src/lib/solve_LINEQ.f90:15-24) -> hecmw_solve -> GPU

The comparison is the reference's own regression harness, examples/test_FrontISTR.rb, restated: `read_log` (:148-223)
parses the max / min summaries of 0.log in both of the formats the tree holds (the *_correct.log files are from an older
version: "Global Summary :Max/Min"), `compare_log` (:225-) accepts |delta| <= 1e-4 absolute on every maximum and minimum.

Decks are the committed copies under tests/golden/decks/ (/root/reference does not exist on the GPU box).  Two
work-arounds of SURVEY.md section 0 are applied to the COPY of the control file, never to the deck: `!RESTART,
FREQUENCY=100000` (+ a !RESTART entry in hecmw_ctrl.dat) because fstr_solve_NLGEOM.f90:204 divides by restart_nout = 0
under flang, and the `!WRITE,VISUAL` / `!VISUAL` block is dropped because the visualizer raises SIGFPE in this build.
"""
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "_ref")
DECKS = os.path.join(os.path.dirname(HERE), "tests", "golden", "decks")
THRESHOLD = 1e-4            # examples/test_FrontISTR.rb:10


def exe(name):
    return os.path.join(REFDIR, name)


def have(name):
    return os.path.exists(exe(name))


def _solver_card(cnt, method=None, precond=None, iterlog=None, timelog=None, line1=None):
    """Rewrite header parameters of the !SOLVER card (fstr_ctrl_common.f90:69-169); the two data lines stay."""
    lines = cnt.split("\n")
    for i, l in enumerate(lines):
        if l.upper().startswith("!SOLVER"):
            for key, val in (("METHOD", method), ("PRECOND", precond), ("ITERLOG", iterlog), ("TIMELOG", timelog)):
                if val is None:
                    continue
                if re.search(key + r"\s*=", l, re.I):
                    l = re.sub(key + r"\s*=\s*\w+", "%s=%s" % (key, val), l, flags=re.I)
                else:
                    l += ",%s=%s" % (key, val)
            lines[i] = l
            if line1 is not None:
                lines[i + 1] = line1
    return "\n".join(lines)


def prepare(deck, workdir, mesh, cnt, **solver):
    """Copy mesh + control file of tests/golden/decks/<deck> into workdir with the two work-arounds and, optionally, another
    METHOD / PRECOND on the !SOLVER card."""
    src = os.path.join(DECKS, deck)
    shutil.copy(os.path.join(src, mesh), os.path.join(workdir, mesh))
    with open(os.path.join(src, cnt)) as fh:
        s = fh.read()
    # the visualizer block: from `!WRITE,VISUAL` lines and from the first `!VISUAL` card to `!END`
    s = "\n".join(l for l in s.split("\n") if not re.match(r"\s*!WRITE\s*,\s*VISUAL", l, re.I))
    m = re.search(r"^\s*!VISUAL", s, re.I | re.M)
    if m:
        s = s[:m.start()].rstrip("\n") + "\n!END\n"
    s = re.sub(r"^(\s*!SOLVER)", "!RESTART, FREQUENCY=100000\n\\1", s, count=1, flags=re.I | re.M)
    if solver:
        s = _solver_card(s, **solver)
    with open(os.path.join(workdir, cnt), "w") as fh:
        fh.write(s)
    with open(os.path.join(workdir, "hecmw_ctrl.dat"), "w") as fh:
        fh.write("!MESH, NAME=fstrMSH,TYPE=HECMW-ENTIRE\n %s\n!CONTROL,NAME=fstrCNT\n %s\n"
                 "!RESULT,NAME=fstrRES,IO=OUT\n out.res\n!RESTART,NAME=restart_out,IO=OUT\n out.restart\n" % (mesh, cnt))


def run(binary, workdir, threads=1, env=None, timeout=600):
    e = dict(os.environ)
    e["OMP_NUM_THREADS"] = str(threads)
    if env:
        e.update(env)
    r = subprocess.run([exe(binary)], cwd=workdir, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    out = {"returncode": r.returncode, "stdout": r.stdout}
    p = os.path.join(workdir, "0.log")
    out["log"] = read_log(p) if os.path.exists(p) else []
    out["heat"] = read_heat_log(p) if os.path.exists(p) else []
    p = os.path.join(workdir, "FSTR.sta")
    out["sta"] = read_sta(p) if os.path.exists(p) else []
    return out


def run_deck(binary, deck, mesh, cnt, threads=1, env=None, keep=None, **solver):
    d = keep or tempfile.mkdtemp(prefix="fistr1_")
    try:
        prepare(deck, d, mesh, cnt, **solver)
        return run(binary, d, threads=threads, env=env)
    finally:
        if keep is None:
            shutil.rmtree(d, ignore_errors=True)


def _to_float(s):
    """'4.5412-317' is accepted like test_FrontISTR.rb:140-146 does."""
    m = re.match(r"^([-+]?[\d.]+)([-+]\d+)$", s)
    return float(m.group(1) + "E" + m.group(2)) if m else float(s)


def read_log(path):
    """-> list of steps, each {'Node': {name: (max, min)}, 'Element': {...}} from the GLOBAL summaries, either format."""
    steps, cur, sect, newfmt = [], None, None, False
    with open(path, errors="replace") as fh:
        for line in fh:
            if "Global Summary :Max/Min" in line or "Global Summary @Node" in line:
                cur = {"Node": {}, "Element": {}}
                steps.append(cur)
                sect, newfmt = cur["Node"], "@Node" in line
            elif cur is not None and sect is cur["Node"] and ("@Element :Max/Min" in line or "Global Summary @Element" in line):
                sect = cur["Element"]
            elif sect is not None and "//" in line:
                a = line.split()
                key = a[0].replace("//", "")
                if not newfmt:
                    key = key.replace("13", "31")
                sect[key] = (_to_float(a[1]), _to_float(a[3] if newfmt else a[2]))
            else:
                sect = None
    return steps


def compare_step(actual, correct, threshold=THRESHOLD):
    """test_FrontISTR.rb compare_item: every key both logs hold, max and min within the threshold.  -> list of mismatches"""
    bad = []
    for part in ("Node", "Element"):
        for k, v in actual[part].items():
            c = correct[part].get(k)
            if c is None:
                continue
            for j, what in ((0, "max"), (1, "min")):
                if not abs(c[j] - v[j]) <= threshold:
                    bad.append((part, k, what, v[j], c[j]))
    return bad


def read_heat_log(path):
    """Heat analyses: the `Maximum Temperature` / `Minimum Temperature` lines of 0.log, what examples/heat/test_heat_sub.sh
    (print_u) extracts -> [('Maximum', t), ('Minimum', t), ...] over the printed steps."""
    out = []
    with open(path, errors="replace") as fh:
        for line in fh:
            m = re.match(r"\s*(Maximum|Minimum) Temperature\s*:\s*(\S+)", line)
            if m:
                out.append((m.group(1), _to_float(m.group(2))))
    return out


def heat_matches(got, want):
    """the logs print temperatures with three decimals: 1e-3 absolute (+ 1e-6 relative for the large ones)"""
    return len(got) == len(want) > 0 and all(a[0] == b[0] and abs(a[1] - b[1]) <= 1e-3 + 1e-6 * abs(b[1]) for a, b in zip(got, want))


def read_sta(path):
    """FSTR.sta rows -> list of (step, substep, status, newton_iterations, message)."""
    rows = []
    with open(path, errors="replace") as fh:
        for line in fh:
            m = re.match(r"\s*(\d+)\s+(\d+)\s*\|\s*(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+\S+\s+\S+\s+\S+\s*\|(.*)", line)
            if m:
                rows.append((int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(6)), m.group(7).strip()))
    return rows


STEP_LINE = re.compile(r"sub_step=|time increment is|State has been restored|Fail to Converge|Number of substeps reached|^\s*iter:\s")


def step_lines(stdout):
    """The lines of fistr1's stdout that describe the incrementation: sub-step headers (time, increment), increment changes,
    cutbacks ('State has been restored'), Newton iterations ('iter: k, residual: r, disp.corr.: d')."""
    return [l.rstrip() for l in stdout.split("\n") if STEP_LINE.search(l)]


def compare_step_lines(got, want, rtol=1e-3, floor=1e-9):
    """Same sequence of lines; numbers within rtol (or below `floor` on both sides: converged residuals are rounding noise)."""
    bad = []
    if len(got) != len(want):
        return [("count", len(got), len(want))]
    num = re.compile(r"[-+]?\d+\.\d+(?:E[-+]\d+)?|[-+]?\d+")
    for a, b in zip(got, want):
        if num.sub("#", a).split() != num.sub("#", b).split():
            bad.append((a, b))
            continue
        for x, y in zip(num.findall(a), num.findall(b)):
            fx, fy = float(x), float(y)
            if abs(fx - fy) > rtol * max(abs(fx), abs(fy)) and max(abs(fx), abs(fy)) > floor:
                bad.append((a, b))
                break
    return bad


def run_restart_pair(binary, deck_dir, workdir, threads=1, env=None):
    """examples/static/restart2/case02_resume as its readme runs it: the first analysis (C3D8beam.cnt, `!RESTART, FREQUENCY=1`,
    SUBSTEPS=3) stops at its third sub-step and leaves a restart file; the second (C3D8beam_res.cnt, `!RESTART, FREQUENCY=-1`)
    resumes from sub-step 4.  deck_dir holds C3D8beam.msh and the two control files.  -> (first run, resumed run)"""
    for f in ("C3D8beam.msh", "C3D8beam.cnt", "C3D8beam_res.cnt"):
        shutil.copy(os.path.join(deck_dir, f), os.path.join(workdir, f))
    outs = []
    for cnt in ("C3D8beam.cnt", "C3D8beam_res.cnt"):
        with open(os.path.join(workdir, "hecmw_ctrl.dat"), "w") as fh:
            fh.write("!MESH, NAME=fstrMSH,TYPE=HECMW-ENTIRE\n C3D8beam.msh\n!CONTROL,NAME=fstrCNT\n %s\n"
                     "!RESTART,NAME=restart_out,IO=INOUT\n C3D8beam.restart\n!RESULT,NAME=fstrRES,IO=OUT\n C3D8beam.res\n" % cnt)
        for stale in ("0.log", "FSTR.sta"):
            if os.path.exists(os.path.join(workdir, stale)):
                os.remove(os.path.join(workdir, stale))
        outs.append(run(binary, workdir, threads=threads, env=env))
    return outs
