"""O2 triplet / cc-pVDZ UHF: per-pass energy, rms and the largest splitting inside the (by symmetry exactly) degenerate pi pairs of the alpha
orbital energies - zero as long as the run sits on the symmetric determinant: tools/o2_trace_dump.py out.txt [npass]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system

m = load_system("oxygen", "cc-pVDZ")
s = q.System(m)
st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 400
with open(sys.argv[1], "w") as f:
    for k in range(npass):
        e, rms = st.iterate()
        w = np.sort(st.orbital_energies(0))
        d = np.diff(w)
        split = d[d < 1e-4].max() if (d < 1e-4).any() else -1.0
        f.write("%d %.15e %.6e %.3e %d\n" % (k, e, rms, split, int((d < 1e-4).sum())))
        if rms / 2.0 < 1e-10:
            break
st.close(); s.close()
