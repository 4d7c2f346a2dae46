/*
 * qchem_hip.h - C ABI of libqchem_hip.so, the MI355X (gfx950) Hartree-Fock hot path that sits behind the public API
 * of qchem-rs's `core` crate.
 *
 * The reference has no FFI of its own; its boundary is the Rust API
 *     core::hf::restricted_hartree_fock(&MolecularSystem, &HartreeFockConfig) -> Option<RestrictedHartreeFockOutput>
 *     core::hf::unrestricted_hartree_fock(...)                                 (core/src/hf/rhf.rs:32, uhf.rs:36)
 * plus the `molint` free functions those drivers call (rhf.rs:41-45, uhf.rs:52-55).  Every entry point below names
 * the reference interface it replaces; INTEGRATION.md shows the `extern "C"` block and the safe wrapper a maintainer
 * would add to `core/src/hf/` to route the two drivers through this library.
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices are n x n, f64, row-major (the ones crossing the boundary are
 *     symmetric, so nalgebra's column-major DMatrix reads them unchanged); coordinates in bohr.
 *   - every call returns a status: QC_OK, QC_NOT_CONVERGED (the reference's `None`), QC_DIIS_SINGULAR (the
 *     reference's `expect("DIIS failed")` panic), or a negative error.  Nothing throws across the boundary.
 *   - a qc_system handle owns its HIP stream and device buffers; it is not thread-safe; distinct handles may be used
 *     from distinct threads.  No global mutable state.
 *   - nothing is printed unless QC_LOG is set: then the drivers write the reference's per-iteration log line (rhf.rs:90-92) to stderr.
 *   - there is NO CPU fallback: any call that needs the GPU returns QC_ERR_NO_DEVICE when no gfx950 device is visible.
 *     Creating a handle, querying sizes, the one-electron matrices and the work plan are host-only and work anywhere.
 */
#ifndef QCHEM_HIP_H
#define QCHEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    QC_OK = 0,
    QC_NOT_CONVERGED = 1,   /* rhf.rs:106-107 / uhf.rs:165-166 return None */
    QC_DIIS_SINGULAR = 2,   /* rhf.rs:73 expect("DIIS failed") / uhf.rs:95-97 panic */
    QC_EIG_NOT_CONVERGED = 3, /* the Jacobi sweeps of an eigensolve ran out (no counterpart: nalgebra's SymmetricEigen, utils.rs:16,
                               * iterates until it converges; reported instead of continuing on unconverged vectors) */
    QC_ERR_INVALID = -1,
    QC_ERR_NO_DEVICE = -2,
    QC_ERR_HIP = -3,
    QC_ERR_RCCL = -4,
    QC_ERR_UNSUPPORTED = -5
};

typedef struct qc_system qc_system;

/* ---- system model: replaces `&MolecularSystem` (molint::system, main.rs:76-77; members read at rhf.rs:36-37).
 * Atoms: atomic numbers + xyz (bohr).  Shells are *segmented* contracted shells: centre atom index, angular momentum
 * (0..3), pure (1 = real solid harmonics, 0 = Cartesian; ignored for L < 2), primitive count, then the primitives of
 * all shells concatenated in `exponents` / `coefficients` (raw contraction coefficients as in the basis-set file;
 * normalisation is done here).  Inputs are copied. */
int qc_system_create(int natoms, const int32_t *atomic_numbers, const double *xyz,
                     int nshells, const int32_t *shell_atom, const int32_t *shell_L, const int32_t *shell_pure,
                     const int32_t *shell_nprim, const double *exponents, const double *coefficients,
                     qc_system **out);
void qc_system_destroy(qc_system *sys);

int qc_nbasis(const qc_system *sys);              /* system.n_basis(), rhf.rs:37 */
int qc_nelectrons(const qc_system *sys);          /* sum of atom.ordinal, rhf.rs:36 */
int qc_nshells(const qc_system *sys);
int64_t qc_nquartets(const qc_system *sys);       /* unique shell quartets (A>=B, C>=D, AB>=CD), before screening */
double qc_nuclear_repulsion(const qc_system *sys);/* compute_nuclear_repulsion, rhf.rs:110-122 */

/* ---- one-electron matrices: replace molint::overlap / kinetic / nuclear (rhf.rs:41-43).  Host code (round 1);
 * out = caller-allocated n*n doubles. */
int qc_overlap(const qc_system *sys, double *out);
int qc_kinetic(const qc_system *sys, double *out);
int qc_nuclear(const qc_system *sys, double *out);
/* The same three matrices computed on the GPU (qc_one_electron.hip; what the SCF drivers use): which = 0 overlap,
 * 1 kinetic, 2 nuclear attraction; out: n*n doubles on the host.  The three entry points above run on the host and need
 * no device. */
int qc_one_electron_gpu(qc_system *sys, int which, double *out);

/* ---- two-electron integrals: replaces molint::eri (rhf.rs:45, uhf.rs:55).  GPU.  out = n^4 doubles on the host,
 * row-major (i,j,k,l), chemists' notation (ij|kl) - the index order rhf.rs:60-61 reads.  Plumbing/tests only: the SCF
 * drivers below never materialise this tensor. */
int qc_eri_full(qc_system *sys, double *out);

/* ---- Fock build: replaces compute_electronic_hamiltonian (rhf.rs:152-167 incl. the tensor of rhf.rs:58-62;
 * uhf.rs:210-227).  GPU, direct: ERI shell quartets are evaluated and digested on the fly.
 *   RHF:  G = J[D] - 1/2 K[D]          (D carries the factor 2 of rhf.rs:179)
 *   UHF:  Ga = J[Da+Db] - K[Da],  Gb = J[Da+Db] - K[Db]
 * Host pointers, n*n each.  With a communicator attached (qc_comm_init) each rank digests its shard of the quartet
 * list and the partial matrices are summed with one RCCL all-reduce. */
int qc_fock_rhf(qc_system *sys, const double *D, double *G);
int qc_fock_uhf(qc_system *sys, const double *Da, const double *Db, double *Ga, double *Gb);
/* Same with device pointers (inputs/outputs resident in HBM); asynchronous on the handle's stream. */
int qc_fock_rhf_device(qc_system *sys, const double *dD, double *dG);
int qc_fock_uhf_device(qc_system *sys, const double *dDa, const double *dDb, double *dGa, double *dGb);

/* ---- symmetric eigensolver: replaces utils::sorted_eigs (hf/utils.rs:20-36).  GPU.  A: n*n symmetric;
 * V: eigenvectors as columns (V[i*n+k] = component i of vector k); w ascending. */
int qc_sym_eig(qc_system *sys, int n, const double *A, double *V, double *w);
/* Same, started from the eigenvectors V0 of a nearby matrix (the SCF loop's case from its second pass on): GEMM-based
 * eigenvector refinement with a Jacobi fallback.  Result identical to qc_sym_eig up to rounding and rotations inside
 * degenerate subspaces. */
int qc_sym_eig_warm(qc_system *sys, int n, const double *A, const double *V0, double *V, double *w);

/* ---- SCF drivers: replace restricted_hartree_fock (rhf.rs:32-108) / unrestricted_hartree_fock (uhf.rs:36-167). */
typedef struct {
    size_t max_iterations;   /* HartreeFockConfig.max_iterations, hf/mod.rs:11 */
    double epsilon;          /* HartreeFockConfig.epsilon, hf/mod.rs:14 */
    /* extension block - zero-initialise for reference behaviour */
    int32_t n_alpha;         /* UHF only; <= 0 with n_beta <= 0: the reference's rule N/2 each (uhf.rs:43-45) */
    int32_t n_beta;
    int32_t reserved[6];
} qc_hf_config;

typedef struct {
    double *orbital_energies;        /* caller buffer, n doubles (RHF) - RestrictedHartreeFockOutput, rhf.rs:14-24 */
    double *orbital_energies_beta;   /* caller buffer, n doubles (UHF beta; alpha goes to orbital_energies), uhf.rs:15-28 */
    double electronic_energy;
    double nuclear_repulsion;
    size_t iterations;               /* 0-based index of the converging iteration (rhf.rs:101) */
    /* timings of the run, milliseconds (not in the reference; the CLI prints its own wall-clock, main.rs:79-101) */
    double ms_setup, ms_fock_total, ms_linalg_total, ms_total;
    double ms_tuner;                 /* of ms_total: host time of the builds that tuned the stream assignment (first build of a geometry) */
} qc_hf_output;

int qc_scf_rhf(qc_system *sys, const qc_hf_config *cfg, qc_hf_output *out);
int qc_scf_uhf(qc_system *sys, const qc_hf_config *cfg, qc_hf_output *out);

/* ---- the same drivers one loop-body pass at a time, for a host that owns the convergence loop itself (the Rust
 * `core` crate in the north-star design; bench.py times exactly K passes through these).
 *   begin   = everything before the loop: integrals, S^-1/2, Hueckel guess, DIIS windows  (rhf.rs:36-65, uhf.rs:40-78)
 *   iterate = one pass of the body: G, F, error, DIIS, eigensolve, new density, energy, rms (rhf.rs:67-88, uhf.rs:81-137);
 *             returns the electronic energy of rhf.rs:84-85 (uhf.rs:145-153) and the density rms the reference
 *             compares with epsilon (for UHF the averaged value of uhf.rs:137, to be tested as rms/2 < epsilon). */
typedef struct qc_scf_state qc_scf_state;
int qc_scf_begin_rhf(qc_system *sys, qc_scf_state **out);
int qc_scf_begin_uhf(qc_system *sys, int n_alpha, int n_beta, qc_scf_state **out);
int qc_scf_iterate(qc_scf_state *st, double *electronic_energy, double *density_rms);
int qc_scf_orbital_energies(qc_scf_state *st, int spin, double *out_n);
int qc_scf_density(qc_scf_state *st, int spin, double *out_nxn);
/* Set-up matrices of the state, n*n doubles to the host: which = 0 overlap S (rhf.rs:41), 1 core Hamiltonian H = T + V
 * (rhf.rs:48), 2 transformation matrix X = S^-1/2 (compute_transformation_matrix, rhf.rs:124-131). */
int qc_scf_matrix(qc_scf_state *st, int which, double *out_nxn);
/* <S^2> of the current UHF determinant: Sz (Sz + 1) + N_beta - tr(D_alpha S D_beta S), Sz = (N_alpha - N_beta) / 2.
 * The reference leaves open shells as a TODO (uhf.rs:42, main.rs:111); this is the diagnostic that goes with the
 * n_alpha / n_beta extension.  RHF states return 0. */
int qc_scf_spin_square(qc_scf_state *st, double *s2);
int qc_scf_timings(qc_scf_state *st, double *ms_setup, double *ms_fock, double *ms_linalg);
/* The stopping rule of the host's loop, told to the library: "I stop calling qc_scf_iterate once rms < epsilon (RHF, rhf.rs:94) /
 * rms / 2 < epsilon (UHF, uhf.rs:139)".  Optional, and it changes no result.  qc_scf_iterate issues the NEXT pass's Fock build behind
 * the pass it is asked for, before it knows how that pass ends (the host is then off the pass boundary); with the rule known, the
 * kernel that ends a pass evaluates it on the device and empties the build queued behind a converging pass instead of running it for
 * nothing.  A host that goes on regardless gets a regular build.  epsilon = 0 (default): no rule; qc_scf_rhf / qc_scf_uhf set theirs. */
int qc_scf_set_stop_rule(qc_scf_state *st, double epsilon);
/* out[0 .. n) of: ms_setup, ms_fock (builds that contained a tuner run are left out), ms_linalg, builds counted in ms_fock, host ms of
 * the first build's timing passes, passes done, speculative builds consumed, speculative builds discarded, passes whose eigensolve was
 * repeated, trials of the online stream-assignment search so far (handle-wide), 1 once that search has ended */
#define QC_SCF_NCOUNTERS 11
int qc_scf_counters(qc_scf_state *st, double *out, int n);
void qc_scf_end(qc_scf_state *st);

/* ---- Fock mode of the SCF drivers on this handle.  0 (default): direct - quartets are evaluated and digested every pass.
 * 1: stored - the reference's own conventional algorithm with the tensor resident in HBM: molint::eri once (rhf.rs:45),
 * electron_terms (rhf.rs:58-62), then one streaming GEMV per pass (rhf.rs:152-167 / uhf.rs:216-226).  Needs ~18 n^4 bytes
 * during set-up (QC_ERR_UNSUPPORTED if the device cannot hold them); single-GPU only. */
int qc_set_fock_mode(qc_system *sys, int mode);
double qc_scf_tensor_ms(qc_scf_state *st);   /* wall time of the tensor build of a stored-mode state */

/* ---- accumulation of the direct Fock build.  1 (default): every contribution to G is rounded once to a multiple of
 * 2^-S Eh and added as a 64-bit integer - the sum does not depend on the order the GPU serves the adds in, so a build is
 * reproducible bit for bit from run to run and under any stream assignment, and both spins of a UHF build see identical
 * arithmetic, as in the reference (uhf.rs:80-108, 210-227).  (Across DIFFERENT shard layouts - numbers of ranks - the sum of the
 * partial matrices agrees to ~1e-13, not bit for bit: the bra-major kernels first combine the exchange rows of a 64-ket bundle in
 * an f64 LDS buffer, and which kets share a bundle depends on the shard.  Non-finite densities give a NaN matrix, as f64 would.)  S follows the density of the build (a bound on |G| keeps every
 * sum inside 64 bits): 2^-49 Eh = 1.8e-15 for benzene/cc-pVDZ.  0: f64 atomics (last-bit noise from the accumulation
 * order; kept for A/B measurements). */
int qc_set_accumulation(qc_system *sys, int fixed_point);

/* ---- Schwarz screening (the reference's TODO at uhf.rs:49-50: "if we could skip some of them ...").  Once per geometry the
 * GPU evaluates the (ab|ab) quartet of every shell pair; unique quartets with sqrt((ab|ab) (cd|cd)) < tau are left out of
 * the work lists (|(ab|cd)| <= that product).  Default 1e-12; 0 switches it off.  Throughput figures keep counting the
 * enumerated quartets (qc_nquartets). */
int qc_set_schwarz(qc_system *sys, double tau);

/* ---- multi-GPU (not in the reference, which is single-threaded; BASELINE.json north_star).  One process per GPU.
 * Rank 0 calls qc_comm_unique_id, the host distributes the 128 bytes, every rank calls qc_comm_init, which creates
 * an RCCL communicator on the current device and restricts this handle's Fock builds to shard `rank` of `nranks`. */
/* RCCL is bound at run time (dlopen): $QC_RCCL_LIB, else the librccl.so.1 already mapped into the process (torch's bundled
 * copy in a torch process - one RCCL per process), else librccl.so.1 from the loader's search path (/opt/rocm/lib).
 * qc_rccl_info writes "<path> version <code>" of the copy in use. */
int qc_rccl_info(char *buf, size_t len);
int qc_comm_unique_id(uint8_t id[128]);
int qc_comm_init(qc_system *sys, const uint8_t id[128], int rank, int nranks);
/* Shard without a communicator (tests / external reduction): Fock builds then return the PARTIAL matrix. */
int qc_set_shard(qc_system *sys, int rank, int nranks);
/* Host-only view of the work plan: number of quartets and the cost-model flops assigned to a shard. */
int qc_plan_shard(qc_system *sys, int rank, int nranks, int64_t *nquartets, double *flops);
/* The shard's quartets as shell indices (A,B,C,D), 4 ints each; abcd == NULL: returns the count only. */
int qc_plan_shard_quartets(qc_system *sys, int rank, int nranks, int32_t *abcd, int64_t capacity);

/* ---- test hook (host only): one entry of a bra-major ket list - packed (pair | first primitive << 18 | length << 25; lists of systems
 * with fewer than 2^18 stored shell pairs) or a plain pair index - written and read back the way the library does. */
int qc_debug_ket_entry(int ket, int first_primitive, int length, int packed, int32_t out[3]);

/* ---- measurement hooks */
/* Dispatch lanes of this handle: the GPU dispatches at most four kernels at a time - hardware queues sit in pairs on four
 * pipes, and a pipe works on one grid until all its workgroups are launched - so the concurrent launches of a Fock build go to one stream
 * per pipe.  Which of the handle's streams share a pipe is measured when the handle initialises its device state.  nlanes: streams on
 * distinct pipes (4 on an MI355X with GPU_MAX_HW_QUEUES=8; 7 = not measured: QC_NO_LANES, or dispatches are serialised by a profiler);
 * slot_stream[0..6]: side stream behind assignment slot k (the first nlanes are the lanes), slot_stream[7]: 1 if slot 0 shares the pipe of
 * the handle's own stream (its chain then runs on that stream itself). */
int qc_dispatch_lanes(qc_system *sys, int32_t *nlanes, int32_t slot_stream[8]);
/* The assignment of a build's launches to the lanes is refined in instalments of extra builds inside later builds (paid for by use,
 * DESIGN.md 3.2).  A harness that is about to TIME builds ends that search here and now, with the best assignment found so far, so that
 * no instalment falls into its timed region.  Results never depend on the assignment. */
int qc_freeze_assignment(qc_system *sys);
int qc_set_stream(qc_system *sys, void *hip_stream);  /* run on the caller's stream (e.g. torch's current stream) */
int qc_device_ready(void);                             /* QC_OK if a gfx950 device is usable */
/* Algorithmic work of one Fock build on this handle's shard (SURVEY.md 8d model): */
typedef struct {
    int64_t quartets;          /* unique shell quartets enumerated */
    int64_t prim_quartets;     /* primitive quartets */
    double bytes_alg;          /* pair data + 6 D blocks read + 6 F blocks read/write, bytes */
    double flops_alg;          /* Boys + R table + Hermite contraction + digestion, flops */
    int32_t nclasses;          /* kernel launches per build */
    int64_t quartets_enumerated;   /* all unique quartets of the molecule (what the reference visits; all ranks) */
    int64_t quartets_screened_out; /* of those, dropped by the Schwarz bound (all ranks); 0 before the device pass has run */
    double schwarz_tau;            /* threshold in force (0: none) */
} qc_work_stats;
int qc_work_stats_get(qc_system *sys, qc_work_stats *out);
/* Time `reps` Fock builds (RHF digestion of dD) per kernel class with hipEvents on the handle's stream.
 * class_ms: caller buffer of `nclasses` floats (average ms per launch of each class bucket); class_id: (BM << 12) | (LAB << 8) |
 * (LCD << 4) | LGC: Hermite orders of the bra / ket pairs, log2 of the lane-group width, BM = 1 for the bra-major kernels. */
int qc_fock_profile(qc_system *sys, const double *dD, double *dG, int reps, float *class_ms, int32_t *class_id,
                    int64_t *class_quartets, double *class_bytes, double *class_flops, float *total_ms);

/* Same for the launch units of an un-instrumented build.  Units 0..13: kernel qc_fock_tier_kernel<LAB, TIER> gathers every
 * class bucket of bra class LAB with LCD <= 3 (TIER 0) or LCD >= 4 (TIER 1), unit = 2 * LAB + TIER.  Units 14..17: the
 * bra-major kernels qc_fock_bm_kernel<LCD, HI> (ket pair ss / ps, bra class LAB <= 2 / LAB >= 3), unit = 14 + 2 * LCD + HI; unit 18:
 * qc_fock_bm_kernel<2, 0>, p.p kets against p.p / d.s bras.
 * All arrays: QC_PROFILE_UNITS entries. */
#define QC_PROFILE_UNITS 20
/* Shell quartets of this rank's shard per launch unit (QC_PROFILE_UNITS entries; no device work). */
int qc_unit_quartets(qc_system *sys, int64_t *unit_quartets);
int qc_fock_profile_tiers(qc_system *sys, const double *dD, double *dG, int reps, float *unit_ms, int64_t *unit_quartets,
                          double *unit_bytes, double *unit_flops, float *total_ms);

/* Measured ceilings of the device the library runs on (bench.py quotes the roofline against them next to the datasheet peaks;
 * SURVEY.md App. F): a register-resident v_fma_f64 loop over the whole chip (TFLOP/s) and a 1 GiB -> 1 GiB streaming copy with
 * 16-byte accesses (GB/s, bytes read + bytes written).  Harness only: no reference counterpart. */
int qc_measure_peaks(double *fp64_tflops, double *hbm_copy_gbs);

#ifdef __cplusplus
}
#endif
#endif
