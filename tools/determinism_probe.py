"""Two Fock builds of the same density: bitwise equal?  (fixed-point accumulation: yes; f64 atomics: last-bit noise)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system

for mol, basis in (("water", "cc-pVTZ"), ("benzene", "cc-pVDZ")):
    m = load_system(mol, basis)
    rng = np.random.default_rng(0)
    for mode in ("fixed", "f64"):
        s = q.System(m); s.set_accumulation(mode)
        D = rng.standard_normal((s.n, s.n)); D = 0.5 * (D + D.T)
        G = [s.fock_rhf(D) for _ in range(4)]
        s2 = q.System(m); s2.set_accumulation(mode)          # fresh handle: its own stream tuning
        G.append(s2.fock_rhf(D))
        print(mol, basis, mode, "max |dG| between builds: %.2e" % max(np.abs(g - G[0]).max() for g in G[1:]),
              "bitwise equal:", all(np.array_equal(g, G[0]) for g in G[1:]), "sym:", np.array_equal(G[0], G[0].T))
        if mode == "fixed":
            Gf = G[0]
        else:
            print("   fixed vs f64: %.2e (scale %.1f)" % (np.abs(Gf - G[0]).max(), np.abs(Gf).max()))
        s.close(); s2.close()
