"""CPU tests of the host side: loaders, the C ABI surface (load + exported symbols, no compute), the host-only model
(one-electron matrices, quartet plan, sharding) and the loud failure when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, data, load_system


def test_loader_reads_reference_formats():
    import qchem_rs_amd as q
    b = q.BasisSet.load(data("basis", "STO-3G.json"))
    assert [s.L for s in b.elements[8]] == [0, 0, 1]                       # SP shell split into s and p
    assert b.elements[8][1].exponents == b.elements[8][2].exponents
    m = q.MolecularSystem.load(data("mol", "water.json"), b)
    assert [a.ordinal for a in m.atoms] == [1, 8, 1] and m.n_basis() == 7 and m.n_electrons == 10
    b2 = q.BasisSet.load(data("basis", "6-31G_st_st.json"))
    assert q.MolecularSystem.load(data("mol", "benzene.json"), b2).n_basis() == 120   # cartesian d
    b3 = q.BasisSet.load(data("basis", "cc-pVTZ.json"))                    # general contractions split, zeros dropped
    m3 = q.MolecularSystem.load(data("mol", "water.json"), b3)
    assert m3.n_basis() == 58 and m3.n_shells == 22
    with pytest.raises(KeyError):
        q.MolecularSystem.load(data("mol", "chloroform.json"), b3)         # no Cl in the authored cc-pVTZ


def test_library_loads_and_exports_every_declared_symbol():
    import qchem_rs_amd as q
    L = q.lib()
    header = open(os.path.join(ROOT, "include", "qchem_hip.h")).read()
    declared = set(re.findall(r"\b(qc_[a-z0-9_]+)\s*\(", header))
    declared -= {"qc_system", "qc_scf_state"}
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(q.hf.EXPORTS) <= declared


def test_host_model_matches_oracle_without_a_gpu():
    import qchem_rs_amd as q
    from oracle.oracle import Oracle
    m = load_system("water", "cc-pVTZ")
    s, o = q.System(m), Oracle(m)
    assert s.n == 58 and s.n_quartets() == 32131 and s.n_electrons() == 10
    assert abs(s.nuclear_repulsion() - o.nuclear_repulsion()) < 1e-13
    assert np.abs(s.overlap() - o.overlap()).max() < 1e-13
    assert np.abs(s.kinetic() - o.kinetic()).max() < 1e-12
    assert np.abs(s.nuclear() - o.nuclear()).max() < 1e-12
    ws = s.work_stats()
    assert ws.quartets == 32131 and ws.prim_quartets > ws.quartets and ws.bytes_alg > 0 and ws.flops_alg > 0


def test_work_lists_are_built_on_demand_and_in_a_fixed_order():
    """Creating a handle sizes its launch classes only; the work lists are built when somebody asks (or behind the Schwarz pass of the device
    set-up).  Whoever asks first gets the same lists: work statistics straight after creation equal those after a shard round trip, and the
    quartet list of a shard - classes in order, inside a class by descending primitive-quartet count, stable - does not depend on how often
    it is asked for (the orders are stable counting sorts: a repeat is identical element for element)."""
    import qchem_rs_amd as q
    m = load_system("water", "cc-pVTZ")
    s1, s2 = q.System(m), q.System(m)
    ws1 = s1.work_stats()                                   # on demand
    s2.set_shard(1, 3); s2.set_shard(0, 1)                  # through a re-shard
    ws2 = s2.work_stats()
    for f in ("quartets", "prim_quartets", "nclasses", "bytes_alg", "flops_alg"):
        assert getattr(ws1, f) == getattr(ws2, f), f
    a, b = s1.plan_shard_quartets(1, 4), s2.plan_shard_quartets(1, 4)
    assert np.array_equal(a, b) and np.array_equal(a, s1.plan_shard_quartets(1, 4))
    assert s1.work_stats().quartets == ws1.quartets         # the plan calls leave the handle's own lists as they were
    s1.close(); s2.close()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_shard_plan_is_a_partition(nranks):
    import qchem_rs_amd as q
    s = q.System(load_system("water", "cc-pVDZ"))
    seen = set()
    counts, flops = [], []
    for r in range(nranks):
        qs = s.plan_shard_quartets(r, nranks)
        nq, fl = s.plan_shard(r, nranks)
        assert nq == len(qs)
        counts.append(nq); flops.append(fl)
        for a, b, c, d in qs.tolist():
            assert a >= b and c >= d
            key = (a, b, c, d) if (a, b) >= (c, d) else (c, d, a, b)
            assert key not in seen
            seen.add(key)
    assert len(seen) == s.n_quartets() == sum(counts)
    assert max(flops) / min(flops) < 1.25                                  # cost-balanced within 25 %


def test_compute_calls_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import qchem_rs_amd as q
    s = q.System(load_system("hydrogen", "STO-3G"))
    assert not q.device_ready()
    with pytest.raises(q.QcError, match="no CPU fallback"):
        s.fock_rhf(np.eye(2))
    with pytest.raises(q.QcError):
        q.restricted_hartree_fock(s, q.HartreeFockConfig(10, 1e-6))
    with pytest.raises(q.QcError):
        s.eri()


def test_invalid_inputs_are_rejected():
    import qchem_rs_amd as q
    L = q.lib()
    h = ctypes.c_void_p()
    Z = np.array([1], np.int32); xyz = np.zeros(3); one = np.array([0], np.int32)
    bad_L = np.array([5], np.int32); npr = np.array([1], np.int32); e = np.array([1.0]); c = np.array([1.0])
    rc = L.qc_system_create(1, Z, xyz, 1, one, bad_L, one, npr, e, c, ctypes.byref(h))
    assert rc == q.hf.QC_ERR_UNSUPPORTED                                    # h shells: no kernels
    bad_atom = np.array([3], np.int32)
    rc = L.qc_system_create(1, Z, xyz, 1, bad_atom, one, one, npr, e, c, ctypes.byref(h))
    assert rc == q.hf.QC_ERR_INVALID


def test_reciprocal_index_division_is_exact_on_the_kernels_range():
    """qc_fdiv (qc_fock_kernel.h): floor(x / d) as (int)((float(x) + 0.5f) * inv) with inv = v_rcp_f32(d), which may be one ulp
    off either way.  The staging / digestion loops of the column kernels divide indices below n_ab * n_cd <= 3 600 by function
    counts <= 100; the identity is checked here, in IEEE float32 arithmetic, on a range sixteen times that."""
    x = np.arange(0, 65536, dtype=np.int64)
    xf = x.astype(np.float32) + np.float32(0.5)
    for d in range(1, 129):
        inv0 = np.float32(1.0) / np.float32(d)
        for inv in (inv0, np.nextafter(inv0, np.float32(0)), np.nextafter(inv0, np.float32(2))):
            assert np.array_equal((xf * inv).astype(np.int64), x // d), d


def test_ket_list_entries_survive_more_than_2_18_shell_pairs():
    """the device-record builder decoded EVERY ket-list entry as (pair | first primitive << 18 | length << 25), also the plain
    pair indices of lists that were never packed (systems with 2^18 or more stored shell pairs, or ket pairs of more than 127 primitives)
    - a plain index above 2^18 lost its upper bits to a chunk that does not exist.  The decode now follows the list's packing flag."""
    import ctypes as C
    import numpy as np
    import qchem_rs_amd as q
    L = q.lib()
    out = (C.c_int32 * 3)()
    for ket in (0, 5, (1 << 18) - 1, 1 << 18, (1 << 18) + 12345, (1 << 25) + 7, (1 << 30) + 3):
        assert L.qc_debug_ket_entry(ket, 0, 0, 0, out) == 0
        assert list(out) == [ket, 0, 0], (ket, list(out))           # plain entries: whole pair, whatever the index
    for ket, f, n in ((0, 0, 1), (77, 3, 8), ((1 << 18) - 1, 120, 7), (1234, 127, 127)):
        assert L.qc_debug_ket_entry(ket, f, n, 1, out) == 0
        assert list(out) == [ket, f, n]
    assert L.qc_debug_ket_entry(1 << 18, 0, 1, 1, out) == -1        # does not fit a packed entry: the list builder never packs then
