"""Per-pass trace of the O2 triplet UHF on the GPU (energy, rms), to compare eigensolver variants (QC_EIG_JACOBI=1) and the oracle's end point."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
from oracle.oracle import Oracle
m = load_system("oxygen", "cc-pVDZ")
s = q.System(m)
st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
npass = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for k in range(npass):
    e, rms = st.iterate()
    print("%3d %.12f %.3e" % (k, e + s.nuclear_repulsion(), rms / 2))
o = Oracle(m)
for eps in (1e-6, 1e-8):
    r = o.uhf(2000, eps, n_alpha=9, n_beta=7)
    print("oracle eps %g: its %d E %.12f" % (eps, r["iterations"], r["total_energy"]))
