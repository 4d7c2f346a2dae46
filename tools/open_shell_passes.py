"""O2 triplet / cc-pVDZ UHF: passes to convergence and final energies of the generic launch sequence and of the one-workgroup Roothaan
kernel (default; QC_NO_OPEN_SHELL_FUSED=1 = the generic sequence), at the CLI's epsilon and at 1e-10 (each mode in a child process: the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import qchem_rs_amd as q
    from conftest import load_system
    s = q.System(load_system("oxygen", "cc-pVDZ"))
    for eps in (1e-6, 1e-10):
        cfg = q.HartreeFockConfig(1000, eps); cfg.n_alpha, cfg.n_beta = 9, 7
        r = q.unrestricted_hartree_fock(s, cfg)
        print("%-8s eps %.0e: %s" % (sys.argv[1], eps, "not converged in 1000" if r is None else "passes %d  E %.12f" % (r.iterations, r.total_energy())))
    s.close()
else:
    for mode in ("generic", "fused"):
        env = dict(os.environ)
        if mode == "generic": env["QC_NO_OPEN_SHELL_FUSED"] = "1"
        print(subprocess.run([sys.executable, __file__, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env).stdout, end="")
