//! Raw declarations of `include/qchem_hip.h` (the subset a `core::hf` replacement needs).
use std::os::raw::{c_double, c_int, c_void};

#[repr(C)]
pub struct QcSystem { _p: [u8; 0] }
#[repr(C)]
pub struct QcScfState { _p: [u8; 0] }

#[repr(C)]
pub struct QcHfConfig {
    pub max_iterations: usize,
    pub epsilon: f64,
    pub n_alpha: i32,
    pub n_beta: i32,
    pub reserved: [i32; 6],
}

#[repr(C)]
pub struct QcHfOutput {
    pub orbital_energies: *mut c_double,
    pub orbital_energies_beta: *mut c_double,
    pub electronic_energy: f64,
    pub nuclear_repulsion: f64,
    pub iterations: usize,
    pub ms_setup: f64,
    pub ms_fock_total: f64,
    pub ms_linalg_total: f64,
    pub ms_total: f64,
    pub ms_tuner: f64,
}

pub const QC_OK: c_int = 0;
pub const QC_NOT_CONVERGED: c_int = 1;
pub const QC_DIIS_SINGULAR: c_int = 2;
pub const QC_EIG_NOT_CONVERGED: c_int = 3;   // the Jacobi sweeps of an eigensolve ran out: an error, not the reference's `None`

extern "C" {
    pub fn qc_system_create(natoms: c_int, z: *const i32, xyz: *const f64, nshells: c_int, shell_atom: *const i32,
        shell_l: *const i32, shell_pure: *const i32, shell_nprim: *const i32, exponents: *const f64,
        coefficients: *const f64, out: *mut *mut QcSystem) -> c_int;
    pub fn qc_system_destroy(sys: *mut QcSystem);
    pub fn qc_nbasis(sys: *const QcSystem) -> c_int;
    pub fn qc_nuclear_repulsion(sys: *const QcSystem) -> f64;
    pub fn qc_overlap(sys: *const QcSystem, out: *mut f64) -> c_int;
    pub fn qc_kinetic(sys: *const QcSystem, out: *mut f64) -> c_int;
    pub fn qc_nuclear(sys: *const QcSystem, out: *mut f64) -> c_int;
    pub fn qc_one_electron_gpu(sys: *mut QcSystem, which: c_int, out: *mut f64) -> c_int;
    pub fn qc_eri_full(sys: *mut QcSystem, out: *mut f64) -> c_int;
    pub fn qc_fock_rhf(sys: *mut QcSystem, d: *const f64, g: *mut f64) -> c_int;
    pub fn qc_fock_uhf(sys: *mut QcSystem, da: *const f64, db: *const f64, ga: *mut f64, gb: *mut f64) -> c_int;
    pub fn qc_sym_eig(sys: *mut QcSystem, n: c_int, a: *const f64, v: *mut f64, w: *mut f64) -> c_int;
    pub fn qc_scf_rhf(sys: *mut QcSystem, cfg: *const QcHfConfig, out: *mut QcHfOutput) -> c_int;
    pub fn qc_scf_uhf(sys: *mut QcSystem, cfg: *const QcHfConfig, out: *mut QcHfOutput) -> c_int;
    pub fn qc_scf_begin_rhf(sys: *mut QcSystem, out: *mut *mut QcScfState) -> c_int;
    pub fn qc_scf_begin_uhf(sys: *mut QcSystem, n_alpha: c_int, n_beta: c_int, out: *mut *mut QcScfState) -> c_int;
    pub fn qc_scf_iterate(st: *mut QcScfState, electronic_energy: *mut f64, density_rms: *mut f64) -> c_int;
    pub fn qc_scf_orbital_energies(st: *mut QcScfState, spin: c_int, out: *mut f64) -> c_int;
    pub fn qc_scf_density(st: *mut QcScfState, spin: c_int, out: *mut f64) -> c_int;
    pub fn qc_scf_spin_square(st: *mut QcScfState, s2: *mut f64) -> c_int;
    pub fn qc_freeze_assignment(sys: *mut QcSystem) -> c_int;
    pub fn qc_dispatch_lanes(sys: *mut QcSystem, nlanes: *mut i32, slot_stream: *mut i32) -> c_int;
    pub fn qc_scf_set_stop_rule(st: *mut QcScfState, epsilon: f64) -> c_int;
    pub fn qc_scf_counters(st: *mut QcScfState, out: *mut f64, n: c_int) -> c_int;
    pub fn qc_scf_end(st: *mut QcScfState);
    pub fn qc_set_fock_mode(sys: *mut QcSystem, mode: c_int) -> c_int;
    /// 1 (default): exact, order-independent accumulation of G; 0: f64 atomics.
    pub fn qc_set_accumulation(sys: *mut QcSystem, fixed_point: c_int) -> c_int;
    /// Schwarz threshold of the work lists (default 1e-12; 0 = every quartet, uhf.rs:49-50).
    pub fn qc_set_schwarz(sys: *mut QcSystem, tau: f64) -> c_int;
    /// 0 overlap, 1 core Hamiltonian, 2 X = S^-1/2 of a state (rhf.rs:41,48,124-131).
    pub fn qc_scf_matrix(st: *mut QcScfState, which: c_int, out: *mut f64) -> c_int;
    /// "<path> version <code>" of the RCCL copy bound at run time.
    pub fn qc_rccl_info(buf: *mut std::os::raw::c_char, len: usize) -> c_int;
    pub fn qc_comm_unique_id(id: *mut u8) -> c_int;
    pub fn qc_comm_init(sys: *mut QcSystem, id: *const u8, rank: c_int, nranks: c_int) -> c_int;
    pub fn qc_set_stream(sys: *mut QcSystem, hip_stream: *mut c_void) -> c_int;
}
