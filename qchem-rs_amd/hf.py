"""Host-side mirror of the reference's `core::hf` API over the C ABI of libqchem_hip.so (include/qchem_hip.h).

Names, argument meaning and error behaviour follow the reference so the parity tests read like its callers:
  HartreeFockConfig{max_iterations, epsilon}            core/src/hf/mod.rs:9-15
  restricted_hartree_fock(system, config) -> Option     core/src/hf/rhf.rs:32-35   (None = not converged, rhf.rs:107)
  unrestricted_hartree_fock(system, config) -> Option   core/src/hf/uhf.rs:36-39
  RestrictedHartreeFockOutput / UnrestrictedHartreeFockOutput incl. total_energy()   rhf.rs:14-30, uhf.rs:15-34
  overlap / kinetic / nuclear / eri                     the molint free functions called at rhf.rs:41-45
A singular DIIS system raises RuntimeError("DIIS failed") where the reference panics (rhf.rs:73, uhf.rs:95-97).

This module is ctypes plumbing only: no arithmetic of the hot path happens in Python, and there is no CPU fallback -
every compute call goes to the HIP library and raises if the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .loader import MolecularSystem

# The Fock build launches its class kernels on several HIP streams.  The chip has four dispatch pipes; with ROCm's default of 4
# hardware queues, streams that land on one queue serialise completely, with 8 the library can pick one queue per pipe for its
# concurrent launches and keep the handle's own stream apart (the library measures which streams share a pipe).
# Must be set before the HIP runtime initialises, hence at import time.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QCHEM_HIP_LIB") or os.path.join(_HERE, "libqchem_hip.so")

QC_OK, QC_NOT_CONVERGED, QC_DIIS_SINGULAR, QC_EIG_NOT_CONVERGED = 0, 1, 2, 3
QC_ERR_INVALID, QC_ERR_NO_DEVICE, QC_ERR_HIP, QC_ERR_RCCL, QC_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
_ERR = {-1: "invalid argument", -2: "no gfx950 device visible (there is no CPU fallback)", -3: "HIP runtime error",
        -4: "RCCL error", -5: "unsupported (angular momentum > f, or n too large for the in-LDS eigensolver)"}

EXPORTS = ["qc_system_create", "qc_system_destroy", "qc_nbasis", "qc_nelectrons", "qc_nshells", "qc_nquartets",
           "qc_nuclear_repulsion", "qc_overlap", "qc_kinetic", "qc_nuclear", "qc_one_electron_gpu", "qc_eri_full", "qc_fock_rhf", "qc_fock_uhf",
           "qc_fock_rhf_device", "qc_fock_uhf_device", "qc_sym_eig", "qc_scf_rhf", "qc_scf_uhf", "qc_comm_unique_id",
           "qc_comm_init", "qc_set_shard", "qc_plan_shard", "qc_set_stream", "qc_device_ready", "qc_work_stats_get",
           "qc_fock_profile", "qc_plan_shard_quartets", "qc_scf_begin_rhf", "qc_scf_begin_uhf", "qc_scf_iterate",
           "qc_scf_orbital_energies", "qc_scf_density", "qc_scf_spin_square", "qc_scf_timings", "qc_scf_end", "qc_fock_profile_tiers", "qc_unit_quartets", "qc_sym_eig_warm", "qc_set_fock_mode", "qc_scf_tensor_ms", "qc_set_accumulation", "qc_set_schwarz", "qc_scf_matrix", "qc_rccl_info", "qc_measure_peaks",
           "qc_scf_set_stop_rule", "qc_scf_counters", "qc_debug_ket_entry", "qc_dispatch_lanes", "qc_freeze_assignment"]


class QcError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("max_iterations", C.c_size_t), ("epsilon", C.c_double), ("n_alpha", C.c_int32), ("n_beta", C.c_int32),
                ("reserved", C.c_int32 * 6)]


class _Output(C.Structure):
    _fields_ = [("orbital_energies", C.POINTER(C.c_double)), ("orbital_energies_beta", C.POINTER(C.c_double)),
                ("electronic_energy", C.c_double), ("nuclear_repulsion", C.c_double), ("iterations", C.c_size_t),
                ("ms_setup", C.c_double), ("ms_fock_total", C.c_double), ("ms_linalg_total", C.c_double),
                ("ms_total", C.c_double), ("ms_tuner", C.c_double)]


class WorkStats(C.Structure):
    _fields_ = [("quartets", C.c_int64), ("prim_quartets", C.c_int64), ("bytes_alg", C.c_double),
                ("flops_alg", C.c_double), ("nclasses", C.c_int32), ("quartets_enumerated", C.c_int64),
                ("quartets_screened_out", C.c_int64), ("schwarz_tau", C.c_double)]


_lib = None
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build_library(force: bool = False) -> str:
    """Compile libqchem_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", csrc])
    return LIB_PATH


def lib():
    """The loaded HIP library.  Raises if it has not been built - the product path never falls back to the CPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QcError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(qchem-rs_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.qc_system_create.argtypes = [C.c_int, _ip, _dp, C.c_int, _ip, _ip, _ip, _ip, _dp, _dp, C.POINTER(vp)]
        L.qc_system_destroy.argtypes = [vp]; L.qc_system_destroy.restype = None
        for f in ("qc_nbasis", "qc_nelectrons", "qc_nshells"):
            getattr(L, f).argtypes = [vp]
        L.qc_nquartets.argtypes = [vp]; L.qc_nquartets.restype = C.c_int64
        L.qc_nuclear_repulsion.argtypes = [vp]; L.qc_nuclear_repulsion.restype = C.c_double
        for f in ("qc_overlap", "qc_kinetic", "qc_nuclear", "qc_eri_full"):
            getattr(L, f).argtypes = [vp, _dp]
        L.qc_one_electron_gpu.argtypes = [vp, C.c_int, _dp]
        L.qc_fock_rhf.argtypes = [vp, _dp, _dp]
        L.qc_fock_uhf.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.qc_fock_rhf_device.argtypes = [vp, vp, vp]
        L.qc_fock_uhf_device.argtypes = [vp, vp, vp, vp, vp]
        L.qc_sym_eig.argtypes = [vp, C.c_int, _dp, _dp, _dp]
        L.qc_sym_eig_warm.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp]
        L.qc_scf_rhf.argtypes = [vp, C.POINTER(_Config), C.POINTER(_Output)]
        L.qc_scf_uhf.argtypes = [vp, C.POINTER(_Config), C.POINTER(_Output)]
        L.qc_comm_unique_id.argtypes = [C.c_char_p]
        L.qc_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.qc_set_shard.argtypes = [vp, C.c_int, C.c_int]
        L.qc_plan_shard.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.qc_plan_shard_quartets.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int64]
        L.qc_set_stream.argtypes = [vp, vp]
        L.qc_device_ready.argtypes = []
        L.qc_work_stats_get.argtypes = [vp, C.POINTER(WorkStats)]
        L.qc_fock_profile.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
        L.qc_fock_profile_tiers.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]
        L.qc_unit_quartets.argtypes = [vp, vp]
        L.qc_set_fock_mode.argtypes = [vp, C.c_int]
        L.qc_set_accumulation.argtypes = [vp, C.c_int]
        L.qc_set_schwarz.argtypes = [vp, C.c_double]
        L.qc_scf_tensor_ms.argtypes = [vp]; L.qc_scf_tensor_ms.restype = C.c_double
        L.qc_scf_begin_rhf.argtypes = [vp, C.POINTER(vp)]
        L.qc_scf_begin_uhf.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.qc_scf_iterate.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.qc_scf_orbital_energies.argtypes = [vp, C.c_int, _dp]
        L.qc_scf_density.argtypes = [vp, C.c_int, _dp]
        L.qc_scf_matrix.argtypes = [vp, C.c_int, _dp]
        L.qc_scf_spin_square.argtypes = [vp, C.POINTER(C.c_double)]
        L.qc_scf_timings.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.qc_scf_end.argtypes = [vp]; L.qc_scf_end.restype = None
        L.qc_scf_set_stop_rule.argtypes = [vp, C.c_double]
        L.qc_scf_counters.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc < 0:
        raise QcError(f"{what}: {_ERR.get(rc, rc)}")
    if rc == QC_EIG_NOT_CONVERGED:
        raise QcError(f"{what}: the Jacobi sweeps of an eigensolve ran out before convergence")
    return rc


def device_ready() -> bool:
    return lib().qc_device_ready() == QC_OK


class System:
    """Owning wrapper of a `qc_system` handle (the `&MolecularSystem` the reference drivers borrow)."""

    def __init__(self, mol: MolecularSystem):
        self.mol = mol
        self._h = C.c_void_p()
        _check(lib().qc_system_create(len(mol.atoms), mol.atomic_numbers(), mol.coordinates(), mol.n_shells, mol.shell_atom,
                                      mol.shell_L, mol.shell_pure, mol.shell_nprim, mol.exponents, mol.coefficients,
                                      C.byref(self._h)), "qc_system_create")
        self.n = lib().qc_nbasis(self._h)

    def close(self):
        if self._h:
            lib().qc_system_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self): return self._h
    def n_basis(self) -> int: return self.n
    def n_electrons(self) -> int: return lib().qc_nelectrons(self._h)
    def n_quartets(self) -> int: return lib().qc_nquartets(self._h)
    def nuclear_repulsion(self) -> float: return lib().qc_nuclear_repulsion(self._h)

    def _mat(self, fn):
        M = np.zeros((self.n, self.n)); _check(getattr(lib(), fn)(self._h, M), fn); return M

    def overlap(self): return self._mat("qc_overlap")
    def kinetic(self): return self._mat("qc_kinetic")
    def nuclear(self): return self._mat("qc_nuclear")

    def one_electron_gpu(self, which: int):
        """S (0), T (1) or V (2) computed by the GPU kernels the SCF drivers use (qc_one_electron.hip)."""
        M = np.zeros((self.n, self.n))
        _check(lib().qc_one_electron_gpu(self._h, which, M.reshape(-1)), "qc_one_electron_gpu")
        return M

    def eri(self):
        I = np.zeros((self.n,) * 4); _check(lib().qc_eri_full(self._h, I.reshape(-1)), "qc_eri_full"); return I

    def fock_rhf(self, D):
        G = np.zeros((self.n, self.n))
        _check(lib().qc_fock_rhf(self._h, np.ascontiguousarray(D, np.float64), G), "qc_fock_rhf"); return G

    def fock_uhf(self, Da, Db):
        Ga, Gb = np.zeros((self.n, self.n)), np.zeros((self.n, self.n))
        _check(lib().qc_fock_uhf(self._h, np.ascontiguousarray(Da, np.float64), np.ascontiguousarray(Db, np.float64), Ga, Gb),
               "qc_fock_uhf")
        return Ga, Gb

    def sym_eig(self, A):
        A = np.ascontiguousarray(A, np.float64); n = A.shape[0]
        V = np.zeros((n, n)); w = np.zeros(n)
        _check(lib().qc_sym_eig(self._h, n, A, V, w), "qc_sym_eig"); return V, w

    def sym_eig_warm(self, A, V0):
        A = np.ascontiguousarray(A, np.float64); n = A.shape[0]
        V = np.zeros((n, n)); w = np.zeros(n)
        _check(lib().qc_sym_eig_warm(self._h, n, A, np.ascontiguousarray(V0, np.float64), V, w), "qc_sym_eig_warm"); return V, w

    def set_fock_mode(self, mode: str):
        """'direct' (default) or 'stored' (the reference's conventional algorithm, tensor resident in HBM)."""
        _check(lib().qc_set_fock_mode(self._h, {"direct": 0, "stored": 1}[mode]), "qc_set_fock_mode")

    def set_accumulation(self, mode: str):
        """'fixed' (default: order-independent 64-bit fixed-point sums, bitwise reproducible) or 'f64' (atomics)."""
        _check(lib().qc_set_accumulation(self._h, {"fixed": 1, "f64": 0}[mode]), "qc_set_accumulation")

    def set_schwarz(self, tau: float):
        """Schwarz threshold of the work lists (default 1e-12; 0 = every quartet, like the reference)."""
        _check(lib().qc_set_schwarz(self._h, float(tau)), "qc_set_schwarz")

    def set_shard(self, rank, nranks): _check(lib().qc_set_shard(self._h, rank, nranks), "qc_set_shard")

    def plan_shard(self, rank, nranks):
        nq, fl = C.c_int64(), C.c_double()
        _check(lib().qc_plan_shard(self._h, rank, nranks, C.byref(nq), C.byref(fl)), "qc_plan_shard")
        return nq.value, fl.value

    def plan_shard_quartets(self, rank, nranks):
        k = _check(lib().qc_plan_shard_quartets(self._h, rank, nranks, None, 0), "qc_plan_shard_quartets")
        out = np.zeros((k, 4), np.int32)
        _check(lib().qc_plan_shard_quartets(self._h, rank, nranks, out.ctypes.data_as(C.c_void_p), k), "qc_plan_shard_quartets")
        return out

    def comm_init(self, uid: bytes, rank: int, nranks: int):
        _check(lib().qc_comm_init(self._h, uid, rank, nranks), "qc_comm_init")

    def freeze_assignment(self):
        """End the search for the stream assignment of the build's launches with what it has found (before timing builds)."""
        _check(lib().qc_freeze_assignment(self._h), "qc_freeze_assignment")

    def dispatch_lanes(self):
        """(number of dispatch lanes, side stream behind every assignment slot, whether slot 0 is the pipe of the handle's own stream)."""
        n = C.c_int32(); sl = (C.c_int32 * 8)()
        _check(lib().qc_dispatch_lanes(self._h, C.byref(n), sl), "qc_dispatch_lanes")
        return n.value, [int(x) for x in sl[:7]], bool(sl[7])

    def set_stream(self, stream_ptr: int): _check(lib().qc_set_stream(self._h, C.c_void_p(stream_ptr)), "qc_set_stream")

    def unit_quartets(self):
        """Shell quartets of this rank's shard per launch unit (see unit_name)."""
        nq = np.zeros(PROFILE_UNITS, np.int64)
        _check(lib().qc_unit_quartets(self._h, nq.ctypes.data_as(C.c_void_p)), "qc_unit_quartets"); return nq

    def work_stats(self) -> WorkStats:
        ws = WorkStats(); _check(lib().qc_work_stats_get(self._h, C.byref(ws)), "qc_work_stats_get"); return ws

    def fock_rhf_device(self, dD_ptr: int, dG_ptr: int):
        _check(lib().qc_fock_rhf_device(self._h, C.c_void_p(dD_ptr), C.c_void_p(dG_ptr)), "qc_fock_rhf_device")

    def fock_profile(self, dD_ptr: int, dG_ptr: int, reps: int):
        k = self.work_stats().nclasses
        ms = np.zeros(k, np.float32); cid = np.zeros(k, np.int32); nq = np.zeros(k, np.int64)
        by = np.zeros(k); fl = np.zeros(k); tot = C.c_float()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(lib().qc_fock_profile(self._h, C.c_void_p(dD_ptr), C.c_void_p(dG_ptr), reps, p(ms), p(cid), p(nq), p(by), p(fl),
                                     C.cast(C.byref(tot), C.c_void_p)), "qc_fock_profile")
        return dict(class_ms=ms, class_id=cid, quartets=nq, bytes=by, flops=fl, total_ms=tot.value)


PROFILE_UNITS = 20   # QC_PROFILE_UNITS in include/qchem_hip.h


def unit_name(u: int) -> str:
    """Kernel instantiation behind launch unit `u` of qc_fock_profile_tiers.  (Inside a build some units share a launch - the wide-ket
    buckets of several bra classes, the ss-ket / high-bra with the ps-ket / low-bra bundles, DESIGN.md 3.1 - and are then reported under
    the unit of the group's last member: its name here, the merged kernel qc_fock_tier1_low_kernel<V> / qc_fock_bm_kernel<3, 0> in a trace.)"""
    return "qc_fock_tier_kernel<%d, %d>" % (u // 2, u % 2) if u < 14 else "qc_fock_bm_kernel<%d, %d>" % ((u - 14) // 2, (u - 14) % 2)


def _fock_profile_tiers(self, dD_ptr: int, dG_ptr: int, reps: int):
    ms = np.zeros(PROFILE_UNITS, np.float32); nq = np.zeros(PROFILE_UNITS, np.int64); by = np.zeros(PROFILE_UNITS); fl = np.zeros(PROFILE_UNITS); tot = C.c_float()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _check(lib().qc_fock_profile_tiers(self._h, C.c_void_p(dD_ptr), C.c_void_p(dG_ptr), reps, p(ms), p(nq), p(by), p(fl),
                                       C.cast(C.byref(tot), C.c_void_p)), "qc_fock_profile_tiers")
    return dict(unit_ms=ms, quartets=nq, bytes=by, flops=fl, total_ms=tot.value)


System.fock_profile_tiers = _fock_profile_tiers


class ScfStepper:
    """One loop-body pass per call (`qc_scf_begin_* / qc_scf_iterate / qc_scf_end`): what a host that owns the
    convergence loop binds, and what bench.py times."""

    def __init__(self, system: "System", uhf: bool = False, n_alpha: int = 0, n_beta: int = 0, stop_rule: float = 0.0):
        """stop_rule > 0: the caller's loop ends once the reference's test holds at that epsilon (rhf.rs:94 / uhf.rs:139); told to the
        library so that the Fock build it queues behind a converging pass is emptied on the device (qc_scf_set_stop_rule)."""
        self.system = system
        self.uhf = uhf
        self._st = C.c_void_p()
        if uhf:
            _check(lib().qc_scf_begin_uhf(system.handle, n_alpha, n_beta, C.byref(self._st)), "qc_scf_begin_uhf")
        else:
            _check(lib().qc_scf_begin_rhf(system.handle, C.byref(self._st)), "qc_scf_begin_rhf")
        if stop_rule > 0.0:
            _check(lib().qc_scf_set_stop_rule(self._st, float(stop_rule)), "qc_scf_set_stop_rule")
        # (the per-pass call is a host's inner loop: the foreign function and its out-parameters are bound once)
        self._iterate = lib().qc_scf_iterate
        self._e, self._r = C.c_double(), C.c_double()
        self._pe, self._pr = C.byref(self._e), C.byref(self._r)

    def iterate(self):
        rc = self._iterate(self._st, self._pe, self._pr)
        if rc != 0:
            if rc == QC_DIIS_SINGULAR:
                raise RuntimeError("DIIS failed")
            _check(rc, "qc_scf_iterate")
        return self._e.value, self._r.value

    def orbital_energies(self, spin=0):
        w = np.zeros(self.system.n); _check(lib().qc_scf_orbital_energies(self._st, spin, w), "qc_scf_orbital_energies"); return w

    def density(self, spin=0):
        D = np.zeros((self.system.n, self.system.n)); _check(lib().qc_scf_density(self._st, spin, D), "qc_scf_density"); return D

    def matrix(self, which: str):
        """Set-up matrix of the state: 'S' (overlap), 'H' (core Hamiltonian) or 'X' (S^-1/2, rhf.rs:124-131)."""
        M = np.zeros((self.system.n, self.system.n))
        _check(lib().qc_scf_matrix(self._st, {"S": 0, "H": 1, "X": 2}[which], M), "qc_scf_matrix"); return M

    def spin_square(self) -> float:
        """<S^2> of the current UHF determinant (0 for RHF)."""
        v = C.c_double(); _check(lib().qc_scf_spin_square(self._st, C.byref(v)), "qc_scf_spin_square"); return v.value

    def tensor_ms(self):
        return lib().qc_scf_tensor_ms(self._st)

    def timings(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _check(lib().qc_scf_timings(self._st, C.byref(a), C.byref(b), C.byref(c)), "qc_scf_timings")
        return dict(setup=a.value, fock=b.value, linalg=c.value)

    def counters(self):
        """ms_setup, ms_fock (builds with a tuner run left out), ms_linalg, builds behind ms_fock, host ms of tuner runs, passes,
        speculative builds consumed / discarded, passes whose eigensolve was repeated."""
        v = (C.c_double * 11)()
        _check(lib().qc_scf_counters(self._st, v, 11), "qc_scf_counters")
        k = ("setup", "fock", "linalg", "builds_timed", "tuner", "passes", "spec_hits", "spec_lost", "redos", "assign_trials", "assign_frozen")
        return dict(zip(k, [float(x) for x in v]))

    def close(self):
        if self._st:
            lib().qc_scf_end(self._st); self._st = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_info() -> str:
    """Path and version of the RCCL library libqchem_hip.so has bound (dlopen; see include/qchem_hip.h)."""
    buf = C.create_string_buffer(512)
    lib().qc_rccl_info(buf, 512)
    return buf.value.decode()


def measure_peaks():
    """(FP64 TFLOP/s of a register-resident v_fma_f64 loop, GB/s of a 1 GiB streaming copy) measured on the current device."""
    f, b = C.c_double(0.0), C.c_double(0.0)
    _check(lib().qc_measure_peaks(C.byref(f), C.byref(b)), "qc_measure_peaks")
    return f.value, b.value


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(lib().qc_comm_unique_id(buf), "qc_comm_unique_id")
    return buf.raw


# ------------------------------------------------------------------ the reference's public API (hf/mod.rs:5-15)
@dataclass
class HartreeFockConfig:
    max_iterations: int = 100      # CLI default, main.rs:33
    epsilon: float = 1e-6          # CLI default, main.rs:36
    n_alpha: int = 0               # extension (uhf only); 0/0 = the reference's N/2 rule (uhf.rs:43-45)
    n_beta: int = 0


@dataclass
class RestrictedHartreeFockOutput:
    orbital_energies: List[float]
    electronic_energy: float
    nuclear_repulsion: float
    iterations: int
    timings_ms: dict = field(default_factory=dict)

    def total_energy(self) -> float:
        return self.electronic_energy + self.nuclear_repulsion


@dataclass
class UnrestrictedHartreeFockOutput:
    orbital_energies_alpha: List[float]
    orbital_energies_beta: List[float]
    electronic_energy: float
    nuclear_repulsion: float
    iterations: int
    timings_ms: dict = field(default_factory=dict)

    def total_energy(self) -> float:
        return self.electronic_energy + self.nuclear_repulsion


def _as_system(system) -> System:
    return system if isinstance(system, System) else System(system)


def _run(fn, system, config, uhf):
    sysh = _as_system(system)
    n = sysh.n
    wa, wb = np.zeros(n), np.zeros(n)
    cfg = _Config(int(config.max_iterations), float(config.epsilon), int(config.n_alpha), int(config.n_beta))
    out = _Output()
    out.orbital_energies = wa.ctypes.data_as(C.POINTER(C.c_double))
    out.orbital_energies_beta = wb.ctypes.data_as(C.POINTER(C.c_double))
    rc = getattr(lib(), fn)(sysh.handle, C.byref(cfg), C.byref(out))
    if rc == QC_DIIS_SINGULAR:
        raise RuntimeError("DIIS failed")                 # rhf.rs:73 / uhf.rs:95-97
    _check(rc, fn)
    if rc == QC_NOT_CONVERGED:
        return None                                       # rhf.rs:106-107
    t = dict(setup=out.ms_setup, fock=out.ms_fock_total, linalg=out.ms_linalg_total, total=out.ms_total, tuner=out.ms_tuner)
    if uhf:
        return UnrestrictedHartreeFockOutput(wa.tolist(), wb.tolist(), out.electronic_energy, out.nuclear_repulsion,
                                             int(out.iterations), t)
    return RestrictedHartreeFockOutput(wa.tolist(), out.electronic_energy, out.nuclear_repulsion, int(out.iterations), t)


def restricted_hartree_fock(system, config: HartreeFockConfig) -> Optional[RestrictedHartreeFockOutput]:
    return _run("qc_scf_rhf", system, config, False)


def unrestricted_hartree_fock(system, config: HartreeFockConfig) -> Optional[UnrestrictedHartreeFockOutput]:
    return _run("qc_scf_uhf", system, config, True)
