"""Set-up cost of a cold handle (QC_SETUP_DEBUG=1 prints the laps): python tools/setup_probe.py mol basis"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
m = load_system(sys.argv[1], sys.argv[2])
for rep in range(2):
    t0 = time.perf_counter(); s = q.System(m); t1 = time.perf_counter()
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10)); t2 = time.perf_counter()
    print("run %d: handle %.2f ms, scf call %.2f ms, timings %s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, out.timings_ms), file=sys.stderr)
    s.close()
