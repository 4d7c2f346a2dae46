"""qchem-rs_amd: MI355X-native Hartree-Fock hot path behind the qchem-rs `core::hf` API."""
from .loader import Atom, BasisSet, MolecularSystem, ShellDef  # noqa: F401
from .hf import (HartreeFockConfig, QcError, RestrictedHartreeFockOutput, ScfStepper, System,  # noqa: F401
                 UnrestrictedHartreeFockOutput, build_library, comm_unique_id, device_ready, lib, measure_peaks, rccl_info,
                 restricted_hartree_fock, unrestricted_hartree_fock)
