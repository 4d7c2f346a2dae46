"""`qchem-hip rhf|uhf ...`: the reference's command line (/root/reference/qchem-cli/src/main.rs) in front of libqchem_hip.so.

Same sub-commands, flags, defaults and printed lines as `qchem-cli` (main.rs:20-62 flags, :98-105 / :143-151 output: three
decimals, Rust's `{:3.3}` / `{:3.3?}` / Duration `{:0.2?}` formats), so a user of the reference finds the contract unchanged.
Two additions, both opt-in:
  * `--json` prints one JSON object with full-precision fields after the reference's lines;
  * `uhf -c/--charge -s/--spin-multiplicity` are honoured (the reference parses and ignores them, main.rs:111 "TODO"):
    with either given, n_alpha / n_beta follow from charge and multiplicity and an `<S^2>` line is added.  With both left at
    0 the behaviour is the reference's N/2 rule (uhf.rs:43-45).
Host-side plumbing only: loaders (loader.py) -> C ABI (hf.py) -> HIP kernels; nothing here computes.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from typing import List, Optional, Sequence, Tuple

from . import hf
from .loader import BasisSet, MolecularSystem


# ---- Rust formatting, as far as main.rs uses it -------------------------------------------------------------------
def fmt_f(x: float) -> str:
    """`{x:3.3}`: three decimals (the minimum width of 3 never binds)."""
    return "%.3f" % x


def fmt_vec(v: Sequence[float]) -> str:
    """`{v:3.3?}` on a Vec<f64>."""
    return "[" + ", ".join(fmt_f(x) for x in v) + "]"


def fmt_duration(seconds: float, precision: Optional[int]) -> str:
    """`{:0.2?}` (precision 2, main.rs:100) and `{:?}` (precision None, main.rs:145) of a std::time::Duration."""
    ns = int(round(seconds * 1e9))
    if ns >= 1_000_000_000:
        value, unit = ns / 1e9, "s"
    elif ns >= 1_000_000:
        value, unit = ns / 1e6, "ms"
    elif ns >= 1_000:
        value, unit = ns / 1e3, "µs"
    else:
        value, unit = float(ns), "ns"
    if precision is not None:
        return "%.*f%s" % (precision, value, unit)
    digits = {"s": 9, "ms": 6, "µs": 3, "ns": 0}[unit]           # Rust prints the exact nanosecond count, zeros trimmed
    text = "%.*f" % (digits, value)
    if "." in text:
        text = text.rstrip("0").rstrip(".")
    return text + unit


# ---- arguments (clap derive of main.rs:9-62) ------------------------------------------------------------------------
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="qchem-hip", description="Hartree-Fock on an MI355X behind the qchem-rs command line")
    p.add_argument("-v", "--verbose", action="store_false", default=True)      # ArgAction::SetFalse, main.rs:16-17
    sub = p.add_subparsers(dest="command", required=True)
    for name in ("rhf", "uhf"):
        s = sub.add_parser(name)
        s.add_argument("-b", "--basis-set", required=True, help="What basis set to use for the hartree fock calculation")
        s.add_argument("-m", "--molecule", required=True, help="A path to the molecule to perform the calculation on")
        if name == "uhf":
            s.add_argument("-c", "--charge", type=int, default=0, help="The charge of the molecule")
            s.add_argument("-s", "--spin-multiplicity", type=int, default=0, help="The spin multiplicity of the molecule")
        s.add_argument("--max-iterations", type=int, default=100)
        s.add_argument("--epsilon", type=float, default=1e-6)
        s.add_argument("--json", action="store_true", help="also print one JSON object with full-precision fields")
    return p


def occupations(n_electrons_neutral: int, charge: int, multiplicity: int) -> Tuple[int, int]:
    """(n_alpha, n_beta) of the extension; multiplicity 0 = the lowest one the electron count allows."""
    n = n_electrons_neutral - charge
    if n <= 0:
        raise ValueError("charge %d leaves no electrons" % charge)
    if multiplicity == 0:
        multiplicity = 1 + (n & 1)
    unpaired = multiplicity - 1
    if unpaired > n or (n - unpaired) % 2:
        raise ValueError("multiplicity %d is impossible with %d electrons" % (multiplicity, n))
    n_beta = (n - unpaired) // 2
    return n_beta + unpaired, n_beta


def _not_converged() -> int:
    print("hartree fock did not converge", file=sys.stderr)                 # panic!, main.rs:106 / :152
    return 101                                                              # exit status of a Rust panic


def run_rhf(args) -> int:
    basis = BasisSet.load(args.basis_set)                                   # main.rs:76
    system = MolecularSystem.load(args.molecule, basis)                     # main.rs:77
    start = time.perf_counter()
    out = hf.restricted_hartree_fock(system, hf.HartreeFockConfig(args.max_iterations, args.epsilon))
    elapsed = time.perf_counter() - start
    if out is None:
        return _not_converged()
    print("hartree fock converged after %d iterations and %s" % (out.iterations, fmt_duration(elapsed, 2)))
    print("electronic energy: " + fmt_f(out.electronic_energy))
    print("nuclear repulsion energy: " + fmt_f(out.nuclear_repulsion))
    print("hartree fock energy: " + fmt_f(out.total_energy()))
    print("orbital energies: " + fmt_vec(out.orbital_energies))
    if args.json:
        print(json.dumps({"method": "rhf", "iterations": out.iterations, "electronic_energy": out.electronic_energy,
                          "nuclear_repulsion": out.nuclear_repulsion, "total_energy": out.total_energy(),
                          "orbital_energies": list(out.orbital_energies), "seconds": elapsed, "timings_ms": out.timings_ms}))
    return 0


def run_uhf(args) -> int:
    basis = BasisSet.load(args.basis_set)
    system = MolecularSystem.load(args.molecule, basis)
    extension = args.charge != 0 or args.spin_multiplicity != 0
    n_alpha = n_beta = 0
    if extension:
        n_alpha, n_beta = occupations(system.n_electrons, args.charge, args.spin_multiplicity)
    start = time.perf_counter()
    s2 = None
    if not extension:
        out = hf.unrestricted_hartree_fock(system, hf.HartreeFockConfig(args.max_iterations, args.epsilon))
    else:
        # the same loop (uhf.rs:82-160) driven pass by pass, so that <S^2> of the final determinant can be read
        handle = hf.System(system)
        st = hf.ScfStepper(handle, uhf=True, n_alpha=n_alpha, n_beta=n_beta)
        out = None
        try:
            for it in range(args.max_iterations + 1):                       # 0..=max_iterations, uhf.rs:82
                e, rms = st.iterate()
                if rms / 2.0 < args.epsilon:                                # uhf.rs:139
                    s2 = st.spin_square()
                    out = hf.UnrestrictedHartreeFockOutput(list(st.orbital_energies(0)), list(st.orbital_energies(1)), e,
                                                           handle.nuclear_repulsion(), it)
                    break
        finally:
            st.close()
            handle.close()
    elapsed = time.perf_counter() - start
    if out is None:
        return _not_converged()
    print("hartree fock converged after %d iterations and %s" % (out.iterations, fmt_duration(elapsed, None)))
    print("electronic energy: " + fmt_f(out.electronic_energy))
    print("nuclear repulsion energy: " + fmt_f(out.nuclear_repulsion))
    print("hartree fock energy: " + fmt_f(out.total_energy()))
    print("orbital energies alpha spin:   " + fmt_vec(out.orbital_energies_alpha))
    print("orbital energies beta spin: " + fmt_vec(out.orbital_energies_beta))
    if s2 is not None:
        print("<S^2>: %s (n_alpha %d, n_beta %d)" % (fmt_f(s2), n_alpha, n_beta))
    if args.json:
        print(json.dumps({"method": "uhf", "iterations": out.iterations, "electronic_energy": out.electronic_energy,
                          "nuclear_repulsion": out.nuclear_repulsion, "total_energy": out.total_energy(),
                          "orbital_energies_alpha": list(out.orbital_energies_alpha),
                          "orbital_energies_beta": list(out.orbital_energies_beta), "n_alpha": n_alpha, "n_beta": n_beta,
                          "spin_square": s2, "seconds": elapsed, "timings_ms": out.timings_ms}))
    return 0


def main(argv: Optional[List[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    return run_rhf(args) if args.command == "rhf" else run_uhf(args)


if __name__ == "__main__":
    sys.exit(main())
