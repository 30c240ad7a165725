"""tests/golden/dump_cube4/: the dump files (`DUMPTYPE=BSR`, Iarray(31)=3) the REFERENCE's hecmw_solve_iterative writes
for the cube4 deck with CG + DIAG: dump_matrix_1_0.{bsr,rhs,sol} (hecmw_matrix_dump.f90).  Build container only."""
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import golden_matrix, load_golden             # noqa: E402
from oracle import refrun                                    # noqa: E402

A = golden_matrix(load_golden("cube4"))
I, R = refrun.default_params(method=1, precond=3)
I[30] = 3                                                    # IDX_I_DUMP = 31: HECMW_MAT_DUMP_TYPE_BSR
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dump_cube4")
os.makedirs(out, exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
    refrun.write_system(fin, 1, A, I, R)
    subprocess.run([os.path.join(refrun.REFDIR, "ref_solve"), fin, fout], cwd=td, check=True, stdout=subprocess.DEVNULL)
    for ext in ("bsr", "rhs", "sol"):
        shutil.copy(os.path.join(td, "dump_matrix_1_0." + ext), os.path.join(out, "dump_matrix_1_0." + ext))
        print(ext, os.path.getsize(os.path.join(out, "dump_matrix_1_0." + ext)))
