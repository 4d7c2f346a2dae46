"""Loaders for the reference's input files (the `molint::basis::BasisSet` / `molint::system::MolecularSystem` API as
the CLI uses it, /root/reference/qchem-cli/src/main.rs:76-77).

`molint` is not in the reference tree, so the accepted formats are read off the shipped data files
(SURVEY.md App. C): MolSSI-BSE schema 0.1 basis JSON and `[{"element": "<Z>", "position": [x,y,z]}]` molecule JSON,
coordinates in bohr.  This is plumbing, not the hot path: it flattens a molecule + basis into the plain arrays the
C ABI (`include/qchem_hip.h: qc_system_create`) takes.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np


@dataclass
class ShellDef:
    """One segmented contracted shell of an element: angular momentum, pure (spherical) flag, primitives."""
    L: int
    pure: bool
    exponents: List[float]
    coefficients: List[float]


class BasisSet:
    """`BasisSet::load(path)` (main.rs:76)."""

    def __init__(self, name: str, elements: Dict[int, List[ShellDef]]):
        self.name = name
        self.elements = elements

    @classmethod
    def load(cls, path) -> "BasisSet":
        with open(path) as f:
            doc = json.load(f)
        elements: Dict[int, List[ShellDef]] = {}
        for z, rec in doc["elements"].items():
            shells: List[ShellDef] = []
            for sh in rec.get("electron_shells", []):
                ftype = sh.get("function_type", "gto")
                exps = [float(e) for e in sh["exponents"]]
                ams = [int(l) for l in sh["angular_momentum"]]
                rows = [[float(c) for c in row] for row in sh["coefficients"]]
                if len(ams) == len(rows):            # plain shell or SP-type shell: one row per L
                    pairs = list(zip(ams, rows))
                elif len(ams) == 1:                  # general contraction: one L, many rows
                    pairs = [(ams[0], row) for row in rows]
                else:
                    raise ValueError(f"{path}: element {z}: cannot pair angular_momentum with coefficients")
                for L, row in pairs:
                    keep = [(e, c) for e, c in zip(exps, row) if c != 0.0]
                    # BSE: "gto" = default harmonic type; only d and higher distinguish cartesian/spherical
                    pure = (L >= 2) and (ftype != "gto_cartesian")
                    shells.append(ShellDef(L, pure, [e for e, _ in keep], [c for _, c in keep]))
            elements[int(z)] = shells
        return cls(doc.get("name", str(path)), elements)


@dataclass
class Atom:
    """`molint::system::Atom` as used at rhf.rs:36,116-117: atomic number + position (bohr)."""
    ordinal: int
    position: Sequence[float]


@dataclass
class MolecularSystem:
    """`MolecularSystem::load(path, &basis)` (main.rs:77) flattened for the C ABI."""
    atoms: List[Atom]
    shell_atom: np.ndarray = field(default=None)
    shell_L: np.ndarray = field(default=None)
    shell_pure: np.ndarray = field(default=None)
    shell_nprim: np.ndarray = field(default=None)
    exponents: np.ndarray = field(default=None)
    coefficients: np.ndarray = field(default=None)

    @classmethod
    def load(cls, path, basis: BasisSet) -> "MolecularSystem":
        with open(path) as f:
            doc = json.load(f)
        atoms = [Atom(int(a["element"]), [float(x) for x in a["position"]]) for a in doc]
        return cls.from_atoms(atoms, basis)

    @classmethod
    def from_atoms(cls, atoms: List[Atom], basis: BasisSet) -> "MolecularSystem":
        sa, sl, sp, sn, ex, co = [], [], [], [], [], []
        for ia, atom in enumerate(atoms):
            if atom.ordinal not in basis.elements:
                raise KeyError(f"basis {basis.name} has no element {atom.ordinal}")
            for sh in basis.elements[atom.ordinal]:
                sa.append(ia); sl.append(sh.L); sp.append(1 if sh.pure else 0); sn.append(len(sh.exponents))
                ex.extend(sh.exponents); co.extend(sh.coefficients)
        return cls(atoms, np.asarray(sa, np.int32), np.asarray(sl, np.int32), np.asarray(sp, np.int32),
                   np.asarray(sn, np.int32), np.asarray(ex, np.float64), np.asarray(co, np.float64))

    # -- the two members the reference's drivers read (rhf.rs:36-37)
    def n_basis(self) -> int:
        n = 0
        for L, pure in zip(self.shell_L, self.shell_pure):
            n += (2 * L + 1) if (pure and L >= 2) else (L + 1) * (L + 2) // 2
        return int(n)

    @property
    def n_electrons(self) -> int:
        return int(sum(a.ordinal for a in self.atoms))

    @property
    def n_shells(self) -> int:
        return int(len(self.shell_L))

    def atomic_numbers(self) -> np.ndarray:
        return np.asarray([a.ordinal for a in self.atoms], np.int32)

    def coordinates(self) -> np.ndarray:
        return np.ascontiguousarray([a.position for a in self.atoms], np.float64).reshape(-1, 3)
