"""ctypes wrapper around oracle/libqc_oracle.so (the CPU restatement in qc_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqc_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "qc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libqc_oracle.so"])
    return _LIB


class _Result(C.Structure):
    _fields_ = [("electronic_energy", C.c_double), ("nuclear_repulsion", C.c_double),
                ("iterations", C.c_long), ("status", C.c_int)]


_lib = None
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_basis_create.restype = C.c_void_p
        L.orc_basis_create.argtypes = [C.c_int, _ip, _dp, C.c_int, _ip, _ip, _ip, _ip, _dp, _dp]
        L.orc_basis_destroy.argtypes = [C.c_void_p]
        for f in ("orc_nbasis", "orc_nshells"):
            getattr(L, f).argtypes = [C.c_void_p]; getattr(L, f).restype = C.c_int
        for f in ("orc_shell_offset", "orc_shell_nfunc", "orc_shell_L", "orc_shell_nprim"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]; getattr(L, f).restype = C.c_int
        for f in ("orc_overlap", "orc_kinetic", "orc_nuclear", "orc_eri_full"):
            getattr(L, f).argtypes = [C.c_void_p, _dp]; getattr(L, f).restype = None
        L.orc_eri_full_strided.argtypes = [C.c_void_p, _dp, C.c_long, C.c_long]; L.orc_eri_full_strided.restype = C.c_long
        L.orc_eri_full_strided_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int]; L.orc_eri_full_strided_mt.restype = C.c_long
        L.orc_n_unique_quartets.argtypes = [C.c_void_p]; L.orc_n_unique_quartets.restype = C.c_long
        L.orc_eri_shell_quartet.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_nuclear_repulsion.argtypes = [C.c_void_p]; L.orc_nuclear_repulsion.restype = C.c_double
        L.orc_eigs.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_sorted_eigs.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_transformation_matrix.argtypes = [C.c_int, _dp, _dp]
        L.orc_antisym_tensor.argtypes = [C.c_int, _dp, _dp]
        L.orc_g_rhf.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_g_uhf.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.orc_g_rhf_quartets.argtypes = [C.c_void_p, C.c_long, _ip, _dp, _dp]
        L.orc_qr_solve.argtypes = [C.c_int, _dp, _dp, _dp]; L.orc_qr_solve.restype = C.c_int
        L.orc_core_guess.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_boys.argtypes = [C.c_int, C.c_double, _dp]
        L.orc_rhf.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_void_p, _dp, C.POINTER(_Result), C.c_void_p,
                              C.POINTER(C.c_long), C.c_void_p, C.c_void_p]
        L.orc_rhf.restype = C.c_int
        L.orc_uhf.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_int, C.c_int, C.c_void_p, _dp, _dp,
                              C.POINTER(_Result), C.c_void_p, C.c_void_p]
        L.orc_uhf.restype = C.c_int
        L.orc_uhf_traced.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_int, C.c_int, C.c_void_p, _dp, _dp,
                                     C.POINTER(_Result), C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.c_void_p]
        L.orc_uhf_traced.restype = C.c_int
        _lib = L
    return _lib


class Oracle:
    """CPU oracle bound to one molecule + basis (a loader.MolecularSystem)."""

    def __init__(self, system):
        L = lib()
        self.system = system
        self.h = L.orc_basis_create(len(system.atoms), system.atomic_numbers(), system.coordinates(),
                                    system.n_shells, system.shell_atom, system.shell_L, system.shell_pure,
                                    system.shell_nprim, system.exponents, system.coefficients)
        self.n = L.orc_nbasis(self.h)
        assert self.n == system.n_basis()

    def __del__(self):
        try:
            lib().orc_basis_destroy(self.h)
        except Exception:
            pass

    def _mat(self, fn):
        M = np.zeros((self.n, self.n)); getattr(lib(), fn)(self.h, M); return M

    def overlap(self): return self._mat("orc_overlap")
    def kinetic(self): return self._mat("orc_kinetic")
    def nuclear(self): return self._mat("orc_nuclear")
    def nuclear_repulsion(self): return lib().orc_nuclear_repulsion(self.h)

    def eri(self):
        I = np.zeros((self.n,) * 4); lib().orc_eri_full(self.h, I.reshape(-1)); return I

    def eri_strided(self, first, stride):
        I = np.zeros((self.n,) * 4); c = lib().orc_eri_full_strided(self.h, I.reshape(-1), first, stride); return I, c

    def eri_strided_mt(self, first, stride, nthreads, store=True):
        """Every stride-th unique shell quartet on `nthreads` host cores (OpenMP); store=False drops the integrals (timing only)."""
        I = np.zeros((self.n,) * 4) if store else None
        c = lib().orc_eri_full_strided_mt(self.h, None if I is None else I.ctypes.data_as(C.c_void_p), first, stride, nthreads)
        return I, c

    def n_unique_quartets(self): return lib().orc_n_unique_quartets(self.h)

    def shell_table(self):
        L = lib()
        return [(L.orc_shell_offset(self.h, s), L.orc_shell_nfunc(self.h, s), L.orc_shell_L(self.h, s),
                 L.orc_shell_nprim(self.h, s)) for s in range(L.orc_nshells(self.h))]

    def eri_shell_quartet(self, a, b, c, d):
        t = self.shell_table()
        out = np.zeros((t[a][1], t[b][1], t[c][1], t[d][1]))
        lib().orc_eri_shell_quartet(self.h, a, b, c, d, out.reshape(-1)); return out

    def core_guess(self):
        n = self.n
        S, H, X, D = (np.zeros((n, n)) for _ in range(4))
        lib().orc_core_guess(self.h, S, H, X, D); return S, H, X, D

    def g_rhf(self, D, eri):
        n = self.n
        T4 = np.zeros(n ** 4); lib().orc_antisym_tensor(n, np.ascontiguousarray(eri).reshape(-1), T4)
        G = np.zeros((n, n)); lib().orc_g_rhf(n, np.ascontiguousarray(D), T4, G); return G

    def g_rhf_quartets(self, D, abcd):
        """G = J - K/2 from a list of unique shell quartets (int32 array, shape (nq, 4))."""
        G = np.zeros((self.n, self.n))
        abcd = np.ascontiguousarray(abcd, np.int32).reshape(-1)
        lib().orc_g_rhf_quartets(self.h, len(abcd) // 4, abcd, np.ascontiguousarray(D, np.float64), G)
        return G

    def g_uhf(self, D1, D2, eri):
        n = self.n
        G = np.zeros((n, n))
        lib().orc_g_uhf(n, np.ascontiguousarray(D1), np.ascontiguousarray(D2), np.ascontiguousarray(eri).reshape(-1), G)
        return G

    def rhf(self, max_iterations=100, epsilon=1e-6, eri=None, trace=False):
        n = self.n
        w = np.zeros(n); res = _Result(); D = np.zeros((n, n))
        tl = C.c_long(0); te = np.zeros(max_iterations + 2); tr = np.zeros(max_iterations + 2)
        ep = None if eri is None else np.ascontiguousarray(eri).ctypes.data_as(C.c_void_p)
        st = lib().orc_rhf(self.h, max_iterations, epsilon, ep, w, C.byref(res), D.ctypes.data_as(C.c_void_p),
                           C.byref(tl), te.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p))
        out = dict(status=st, electronic_energy=res.electronic_energy, nuclear_repulsion=res.nuclear_repulsion,
                   iterations=res.iterations, orbital_energies=w, density=D,
                   total_energy=res.electronic_energy + res.nuclear_repulsion)
        if trace:
            out["trace_energy"] = te[:tl.value].copy(); out["trace_rms"] = tr[:tl.value].copy()
        return out

    def uhf(self, max_iterations=100, epsilon=1e-6, n_alpha=-1, n_beta=-1, eri=None, trace=False):
        n = self.n
        wa, wb = np.zeros(n), np.zeros(n); res = _Result(); Da, Db = np.zeros((n, n)), np.zeros((n, n))
        ep = None if eri is None else np.ascontiguousarray(eri).ctypes.data_as(C.c_void_p)
        tl = C.c_long(0); te = np.zeros(max_iterations + 2); tr = np.zeros(max_iterations + 2)
        st = lib().orc_uhf_traced(self.h, max_iterations, epsilon, n_alpha, n_beta, ep, wa, wb, C.byref(res),
                                  Da.ctypes.data_as(C.c_void_p), Db.ctypes.data_as(C.c_void_p), C.byref(tl),
                                  te.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p))
        extra = dict(trace_energy=te[:tl.value].copy(), trace_rms=tr[:tl.value].copy()) if trace else {}
        return dict(**extra, status=st, electronic_energy=res.electronic_energy, nuclear_repulsion=res.nuclear_repulsion,
                    iterations=res.iterations, orbital_energies_alpha=wa, orbital_energies_beta=wb,
                    density_alpha=Da, density_beta=Db,
                    total_energy=res.electronic_energy + res.nuclear_repulsion)


def sorted_eigs(A):
    n = A.shape[0]; V = np.zeros((n, n)); w = np.zeros(n)
    lib().orc_sorted_eigs(n, np.ascontiguousarray(A, np.float64), V, w); return V, w


def boys(nmax, x):
    F = np.zeros(nmax + 1); lib().orc_boys(nmax, float(x), F); return F
