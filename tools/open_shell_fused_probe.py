"""O2 triplet / cc-pVDZ UHF (BASELINE config 4): the pass-by-pass trace of the one-workgroup Roothaan kernel (default; QC_NO_OPEN_SHELL_FUSED=1: the generic launch sequence) against
the generic launch sequence's, bit for bit (hex), and the linear-algebra time per pass of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system
m = load_system("oxygen", "cc-pVDZ")
def run(fused, npass=60):
    if fused: os.environ.pop("QC_NO_OPEN_SHELL_FUSED", None)
    else: os.environ["QC_NO_OPEN_SHELL_FUSED"] = "1"
    s = q.System(m)
    warm = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    for _ in range(5): warm.iterate()
    warm.close()
    st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    tr = []
    for k in range(npass):
        e, r = st.iterate(); tr.append((e, r))
        if r / 2 < 1e-10: break
    c = st.counters(); st.close(); s.close()
    return tr, c
# (QC_NO_OPEN_SHELL_FUSED is read once per process: the fused run is a child process)
if len(sys.argv) > 1:
    tr, c = run(sys.argv[1].startswith("fused"))
    for k, (e, r) in enumerate(tr): print("T %d %s %s" % (k, float(e).hex(), float(r).hex()))
    print("C passes %d linalg_ms_per_pass %.4f fock_ms_per_build %.4f" % (len(tr), c["linalg"] / len(tr), c["fock"] / max(1, c["builds_timed"])))
else:
    import subprocess
    out = {}
    for mode in ("generic", "fused", "serial", "fused_serial"):
        env = dict(os.environ)
        if not mode.startswith("fused"): env["QC_NO_OPEN_SHELL_FUSED"] = "1"
        if mode.endswith("serial"): env["QC_NO_SPIN_PARALLEL"] = "1"
        p = subprocess.run([sys.executable, __file__, mode], stdout=subprocess.PIPE, text=True, env=env)
        out[mode] = p.stdout.splitlines()
        print(mode, [l for l in out[mode] if l.startswith("C")])
    a = [l for l in out["generic"] if l.startswith("T")]; b = [l for l in out["serial"] if l.startswith("T")]
    print("spin-parallel against serial: identical lines %d of %d" % (sum(1 for x, y in zip(a, b) if x == y), len(a)))
    b = [l for l in out["fused"] if l.startswith("T")]
    same = sum(1 for x, y in zip(a, b) if x == y)
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
    print("passes generic %d fused %d, identical lines %d, first difference at pass %s" % (len(a), len(b), same, first))
    if first is not None: print(a[first]); print(b[first])
    c = [l for l in out["fused_serial"] if l.startswith("T")]
    print("fused, spins side by side against one after the other: identical lines %d of %d" % (sum(1 for x, y in zip(b, c) if x == y), len(b)))
