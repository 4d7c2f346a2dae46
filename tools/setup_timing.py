#!/usr/bin/env python3
"""Set-up time of an SCF run (one-electron matrices on the GPU, X = S^-1/2, guess) and the GPU vs host one-electron time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qchem_rs_amd as q
for molname, basis in (("water", "cc-pVTZ"), ("benzene", "cc-pVDZ")):
    b = q.BasisSet.load("data/basis/%s.json" % basis)
    m = q.MolecularSystem.load("data/mol/%s.json" % molname, b)
    s = q.System(m)
    s.one_electron_gpu(0)                                   # device init + first launch
    t0 = time.perf_counter(); [s.one_electron_gpu(w) for w in (0, 1, 2)]; tg = time.perf_counter() - t0
    t0 = time.perf_counter(); s.overlap(); s.kinetic(); s.nuclear(); th = time.perf_counter() - t0
    st = q.ScfStepper(s)
    print("%s/%s: S+T+V on the GPU %.2f ms (incl. 3 copies), on the host %.2f ms; qc_scf_begin %.1f ms" % (molname, basis, tg * 1e3, th * 1e3, st.timings()["setup"]))
    st.close(); s.close()
