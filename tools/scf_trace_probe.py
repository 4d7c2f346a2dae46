"""Per-pass trace (energy, density rms) of the GPU SCF, optionally next to the oracle's: tools/scf_trace_probe.py mol basis npass [oracle] [f64]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system

mol, basis, npass = sys.argv[1], sys.argv[2], int(sys.argv[3])
m = load_system(mol, basis)
s = q.System(m)
if "f64" in sys.argv:
    s.set_accumulation("f64")
st = q.ScfStepper(s)
S = st.matrix("S")
tr = []
t0 = time.time()
for k in range(npass):
    e, rms = st.iterate()
    tr.append((e, rms))
print("gpu: %d passes in %.2f s" % (npass, time.time() - t0))
D = st.density(0)
print("tr(DS) = %.9f (N = %d)  max|D| %.3e" % (np.sum(D * S), s.n_electrons(), np.abs(D).max()))
ref = None
if "oracle" in sys.argv:
    from oracle.oracle import Oracle
    o = Oracle(m)
    t0 = time.time()
    ref = o.rhf(npass - 1, 1e-30, trace=True)
    print("oracle: %.1f s" % (time.time() - t0))
for k, (e, rms) in enumerate(tr):
    line = "%3d  E %.12f  rms %.3e" % (k, e, rms)
    if ref is not None and k < len(ref["trace_energy"]):
        line += "   | oracle E %.12f rms %.3e  dE %.2e" % (ref["trace_energy"][k], ref["trace_rms"][k], e - ref["trace_energy"][k])
    print(line)
