#!/usr/bin/env python3
"""O2 triplet / cc-pVDZ UHF (BASELINE config 4): per-iteration time of the step API (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qchem_rs_amd as q
b = q.BasisSet.load("data/basis/cc-pVDZ.json")
m = q.MolecularSystem.load("data/mol/oxygen.json", b)
s = q.System(m)
st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
ts = []
for i in range(30):
    t0 = time.perf_counter(); e, r = st.iterate(); ts.append(time.perf_counter() - t0)
tm = st.timings()
print("O2 triplet/cc-pVDZ UHF n=%d quartets=%d: E_elec %.10f rms %.2e; ms/iter (last 15) %.3f; fock %.3f linalg %.3f (avg ms over 30)" % (
    s.n, s.n_quartets(), e, r, 1e3 * sum(ts[15:]) / 15, tm["fock"] / 30, tm["linalg"] / 30))
st.close(); s.close()
