"""Host-side time stamps of SCF passes (QC_ISSUE_DEBUG=1): where the host's time goes between the end of a pass and the launches of the
next build.  usage: QC_ISSUE_DEBUG=1 python tools/issue_probe.py [mol basis]"""
import os, sys
os.environ.setdefault("QC_ISSUE_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
mol, basis = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("water", "cc-pVTZ")
s = q.System(load_system(mol, basis))
for rep in range(3):
    st = q.ScfStepper(s, stop_rule=1e-10)
    sys.stderr.write("--- run %d\n" % rep)
    for k in range(15):
        e, rms = st.iterate()
    st.close()
s.close()
