"""A/B of the two accumulation modes of the direct Fock build (fixed point, exact and order-independent, vs f64 atomics):
hipEvent time of the build inside SCF passes, and whether repeated builds agree bit for bit."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system

out = {}
for mol, basis in (("water", "cc-pVTZ"), ("benzene", "cc-pVDZ")):
    m = load_system(mol, basis)
    rng = np.random.default_rng(0)
    n = None
    res = {}
    for rep in range(2):
        for mode in ("fixed", "f64"):
            s = q.System(m); s.set_accumulation(mode)
            if n is None:
                n = s.n; D = rng.standard_normal((n, n)); D = 0.5 * (D + D.T)
            G = [s.fock_rhf(D) for _ in range(3)]
            st = q.ScfStepper(s)
            for _ in range(6):
                st.iterate()
            t0 = st.timings(); w0 = time.perf_counter()
            K = 20
            for _ in range(K):
                st.iterate()
            w1 = time.perf_counter(); t1 = st.timings()
            r = res.setdefault(mode, {"fock_ms": [], "iter_ms": []})
            r["fock_ms"].append((t1["fock"] - t0["fock"]) / K); r["iter_ms"].append((w1 - w0) * 1e3 / K)
            r["bitwise_equal_builds"] = bool(all(np.array_equal(g, G[0]) for g in G[1:]))
            r["G"] = G[0]
            st.close(); s.close()
    d = float(np.abs(res["fixed"]["G"] - res["f64"]["G"]).max())
    for mode in res:
        del res[mode]["G"]
    res["max_abs_diff_fixed_vs_f64"] = d
    out["%s/%s" % (mol, basis)] = res
print(json.dumps(out, indent=1))
