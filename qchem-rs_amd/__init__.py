"""qchem-rs_amd: MI355X-native Hartree-Fock hot path behind the qchem-rs `core::hf` API."""
from .loader import Atom, BasisSet, MolecularSystem, ShellDef  # noqa: F401
