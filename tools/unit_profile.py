"""Launch units of a build timed alone: python tools/unit_profile.py <workload> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qchem_rs_amd as q
import bench
mol = bench.load(q, sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = q.System(mol)
rng = np.random.default_rng(0)
D = rng.standard_normal((s.n, s.n)); D = D + D.T
dD = torch.from_numpy(D).cuda(); dG = torch.zeros_like(dD)
s.fock_profile_tiers(dD.data_ptr(), dG.data_ptr(), 1)
tp = s.fock_profile_tiers(dD.data_ptr(), dG.data_ptr(), reps)
for u in range(len(tp["unit_ms"])):
    if tp["quartets"][u] > 0:
        print("%-28s %7d quartets %8.1f us" % (q.hf.unit_name(u), tp["quartets"][u], tp["unit_ms"][u] * 1e3))
print("sum %.1f us, build %.1f us" % (tp["unit_ms"].sum() * 1e3, tp["total_ms"] * 1e3))
