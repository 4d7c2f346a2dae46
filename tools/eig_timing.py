"""Wall time of qc_sym_eig (cold symmetric eigensolve through the C ABI, including its allocations and two PCIe copies) for random
symmetric matrices and for a matrix with degenerate pairs; QC_EIG_JACOBI=1 selects the single-workgroup Jacobi kernels (A/B)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qchem_rs_amd as q
from conftest import load_system

s = q.System(load_system("hydrogen", "STO-3G"))
rng = np.random.default_rng(1)
sizes = [int(a) for a in sys.argv[1:]] or [24, 58, 114, 139, 150, 174, 256]
for n in sizes:
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
    # second matrix: doubly degenerate spectrum
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w0 = np.repeat(np.linspace(-5, 5, (n + 1) // 2), 2)[:n]
    B = (Q * w0) @ Q.T; B = 0.5 * (B + B.T)
    for name, M in (("random", A), ("degenerate pairs", B)):
        V, w = s.sym_eig(M)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            V, w = s.sym_eig(M)
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))                       # (a call allocates and frees its work space: single calls now and then take milliseconds)
        wr = np.linalg.eigvalsh(M)
        print("n=%3d %-17s %.3f ms (median of 10, max %.3f)  |dw| %.1e  orth %.1e  resid %.1e" % (n, name, dt * 1e3, max(ts) * 1e3, np.abs(w - wr).max(), np.abs(V.T @ V - np.eye(n)).max(),
                                                                       np.abs(M @ V - V * w).max()))
