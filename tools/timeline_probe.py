"""Device-side timeline of SCF passes (QC_DEV_TIMELINE=1: every kernel of a pass leaves the clock of its first start and last end; no
profiler in the process).  Three warm-up runs (assignment search, caches), then one run whose passes are printed.
usage: python tools/timeline_probe.py [mol basis]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
mol, basis = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("water", "cc-pVTZ")
s = q.System(load_system(mol, basis))
for rep in range(int(os.environ.get("WARM_RUNS", "12"))):
    st = q.ScfStepper(s, stop_rule=1e-10)
    for k in range(15):
        st.iterate()
    st.close()
s.freeze_assignment()
os.environ["QC_DEV_TIMELINE"] = "1"
for rep in range(2):
    st = q.ScfStepper(s, stop_rule=1e-10)
    sys.stderr.write("--- run %d\n" % rep)
    for k in range(15):
        st.iterate()
    c = st.counters()
    sys.stderr.write("    (events of the same passes: build %.4f ms, linear algebra %.4f ms per pass)\n" % (c["fock"] / max(1.0, c["builds_timed"]), c["linalg"] / 15))
    st.close()
s.close()
