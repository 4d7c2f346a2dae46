"""O2 triplet / cc-pVDZ UHF (BASELINE config 4): run the GPU SCF several times and compare with the oracle - iteration of
convergence, reported energy, variational energy of the converged densities, bitwise reproducibility of the runs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
from oracle.oracle import Oracle

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = load_system("oxygen", "cc-pVDZ")
o = Oracle(m)
ref = o.uhf(2000, 1e-10, n_alpha=9, n_beta=7)
I, H = o.eri(), o.kinetic() + o.nuclear()
evar = lambda A, B: 0.5 * np.sum(A * (2 * H + o.g_uhf(A, B, I))) + 0.5 * np.sum(B * (2 * H + o.g_uhf(B, A, I)))
e_ref = evar(ref["density_alpha"], ref["density_beta"])
print("oracle: status %d its %d E %.12f Evar %.12f" % (ref["status"], ref["iterations"], ref["total_energy"], e_ref + o.nuclear_repulsion()))
prev = None
for r in range(runs):
    s = q.System(m)
    if len(sys.argv) > 2:
        s.set_accumulation(sys.argv[2])
    st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    t0 = time.time()
    trace = []
    for it in range(3000):
        e, rms = st.iterate()
        trace.append((e, rms))
        if rms / 2.0 < 1e-10:
            break
    Da, Db = st.density(0), st.density(1)
    dE = evar(Da, Db) - e_ref
    same = None if prev is None else (trace == prev)
    prev = trace
    print("gpu run %d: its %d rms %.2e E %.12f dE_reported %.2e dEvar %.2e  same-as-previous-run %s  %.1fs" %
          (r, it, rms, e + s.nuclear_repulsion(), e + s.nuclear_repulsion() - ref["total_energy"], dE, same, time.time() - t0))
    st.close(); s.close()
