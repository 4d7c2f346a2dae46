#!/usr/bin/env python3
"""Per-class serial timing of one Fock build (diagnostic; run on the GPU box): python tools/class_profile.py <workload> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qchem_rs_amd as q
import bench
key = sys.argv[1] if len(sys.argv) > 1 else "c6h6_ccpvdz"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mol = bench.load(q, key)
s = q.System(mol)
n = s.n
rng = np.random.default_rng(0)
D = rng.standard_normal((n, n)); D = D + D.T
dD = torch.from_numpy(D).cuda(); dG = torch.zeros_like(dD)
s.fock_profile(dD.data_ptr(), dG.data_ptr(), 1)
pr = s.fock_profile(dD.data_ptr(), dG.data_ptr(), reps)
tp = s.fock_profile_tiers(dD.data_ptr(), dG.data_ptr(), reps)
tot = 0.0
rows = []
for i in range(len(pr["class_ms"])):
    c = int(pr["class_id"][i]); ms = float(pr["class_ms"][i]); tot += ms
    nm = ("bm<%d,%d>" % ((c >> 8) & 15, (c >> 4) & 15)) if c >> 12 else "<%d,%d,%d>" % (c >> 8, (c >> 4) & 15, c & 15)
    rows.append((ms, nm, int(pr["quartets"][i]), float(pr["flops"][i]) / 1e9))
for ms, nm, nq, gf in sorted(rows, reverse=True):
    print("%-10s %8d quartets %8.3f GF %8.4f ms %7.2f TF/s" % (nm, nq, gf, ms, gf / ms if ms > 0 else 0))
print("sum of classes %.3f ms; tiers serial %.3f ms; build %.3f ms" % (tot, float(tp["unit_ms"].sum()), float(tp["total_ms"])))
s.close()
