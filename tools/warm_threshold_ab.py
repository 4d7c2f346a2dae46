"""Per-pass linear-algebra time and redo count of whole SCF runs for different values of QC_EIG_WARM_RMS (set in the environment)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
for mol, basis, uhf in (("water", "cc-pVTZ", False), ("benzene", "cc-pVDZ", False), ("oxygen", "cc-pVDZ", True), ("ethylene", "6-31G_st_st", False)):
    s = q.System(load_system(mol, basis))
    for rep in range(2):
        st = q.ScfStepper(s, uhf=uhf, n_alpha=9 if uhf else 0, n_beta=7 if uhf else 0)
        t0 = time.perf_counter(); k = 0
        for k in range(60):
            e, rms = st.iterate()
            if (rms / 2 if uhf else rms) < 1e-9:
                break
        dt = time.perf_counter() - t0
        tm = st.timings()
        st.close()
    print("%-10s %-12s passes %2d  wall/pass %.3f ms  fock %.3f  linalg %.3f   E %.10f" % (mol, basis, k + 1, dt * 1e3 / (k + 1), tm["fock"] / (k + 1), tm["linalg"] / (k + 1), e))
    s.close()
