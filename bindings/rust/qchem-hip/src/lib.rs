//! Safe wrapper with the shape of qchem-rs's `core::hf` API (`core/src/hf/mod.rs:5-15`, `rhf.rs:11-35`,
//! `uhf.rs:12-39`): the same config struct, the same output structs, `Option` for "not converged", a panic for
//! "DIIS failed".  Input is a flattened description of the molecule and its segmented shells, because the reference's
//! `MolecularSystem` lives in the `molint` crate, which is not part of the reference tree; a `From<&MolecularSystem>`
//! for `FlatSystem` is the one piece a maintainer with `molint` at hand has to add.
//!
//! SOURCE ONLY - not compiled in the image this was written in (no Rust toolchain there).
pub mod ffi;

use std::ptr::null_mut;

/// `core::hf::HartreeFockConfig` (hf/mod.rs:9-15)
pub struct HartreeFockConfig { pub max_iterations: usize, pub epsilon: f64 }

/// rhf.rs:14-24
#[non_exhaustive]
pub struct RestrictedHartreeFockOutput { pub orbital_energies: Vec<f64>, pub electronic_energy: f64, pub nuclear_repulsion: f64, pub iterations: usize }
impl RestrictedHartreeFockOutput { pub fn total_energy(&self) -> f64 { self.electronic_energy + self.nuclear_repulsion } }

/// uhf.rs:15-27
#[non_exhaustive]
pub struct UnrestrictedHartreeFockOutput {
    pub orbital_energies_alpha: Vec<f64>, pub orbital_energies_beta: Vec<f64>,
    pub electronic_energy: f64, pub nuclear_repulsion: f64, pub iterations: usize,
}
impl UnrestrictedHartreeFockOutput { pub fn total_energy(&self) -> f64 { self.electronic_energy + self.nuclear_repulsion } }

/// What `qc_system_create` takes: atoms (Z, bohr), segmented shells (centre, L, pure flag, primitives).
pub struct FlatSystem {
    pub z: Vec<i32>, pub xyz: Vec<f64>,
    pub shell_atom: Vec<i32>, pub shell_l: Vec<i32>, pub shell_pure: Vec<i32>, pub shell_nprim: Vec<i32>,
    pub exponents: Vec<f64>, pub coefficients: Vec<f64>,
}

/// Owning handle: pair data, work lists and device buffers live behind it.  Not `Sync`: one handle, one thread.
pub struct GpuSystem { ptr: *mut ffi::QcSystem }
impl GpuSystem {
    pub fn new(s: &FlatSystem) -> Option<Self> {
        let mut ptr = null_mut();
        let rc = unsafe {
            ffi::qc_system_create(s.z.len() as i32, s.z.as_ptr(), s.xyz.as_ptr(), s.shell_l.len() as i32, s.shell_atom.as_ptr(),
                s.shell_l.as_ptr(), s.shell_pure.as_ptr(), s.shell_nprim.as_ptr(), s.exponents.as_ptr(), s.coefficients.as_ptr(), &mut ptr)
        };
        if rc == ffi::QC_OK { Some(GpuSystem { ptr }) } else { None }
    }
    pub fn n_basis(&self) -> usize { unsafe { ffi::qc_nbasis(self.ptr) as usize } }
    pub fn as_ptr(&self) -> *mut ffi::QcSystem { self.ptr }
}
impl Drop for GpuSystem { fn drop(&mut self) { unsafe { ffi::qc_system_destroy(self.ptr) } } }

fn config(c: &HartreeFockConfig) -> ffi::QcHfConfig {
    ffi::QcHfConfig { max_iterations: c.max_iterations, epsilon: c.epsilon, n_alpha: 0, n_beta: 0, reserved: [0; 6] }
}

/// `core::hf::restricted_hartree_fock` (rhf.rs:32-108) on the GPU.
pub fn restricted_hartree_fock(system: &FlatSystem, config_in: &HartreeFockConfig) -> Option<RestrictedHartreeFockOutput> {
    let h = GpuSystem::new(system)?;
    let mut eps = vec![0.0f64; h.n_basis()];
    let cfg = config(config_in);
    let mut out: ffi::QcHfOutput = unsafe { std::mem::zeroed() };
    out.orbital_energies = eps.as_mut_ptr();
    match unsafe { ffi::qc_scf_rhf(h.ptr, &cfg, &mut out) } {
        ffi::QC_OK => Some(RestrictedHartreeFockOutput { orbital_energies: eps, electronic_energy: out.electronic_energy,
                                                        nuclear_repulsion: out.nuclear_repulsion, iterations: out.iterations }),
        ffi::QC_NOT_CONVERGED => None,                       // rhf.rs:106-107
        ffi::QC_DIIS_SINGULAR => panic!("DIIS failed"),      // rhf.rs:73
        e => panic!("qchem_hip error {e}"),
    }
}

/// `core::hf::unrestricted_hartree_fock` (uhf.rs:36-167) on the GPU.
pub fn unrestricted_hartree_fock(system: &FlatSystem, config_in: &HartreeFockConfig) -> Option<UnrestrictedHartreeFockOutput> {
    let h = GpuSystem::new(system)?;
    let n = h.n_basis();
    let (mut ea, mut eb) = (vec![0.0f64; n], vec![0.0f64; n]);
    let cfg = config(config_in);
    let mut out: ffi::QcHfOutput = unsafe { std::mem::zeroed() };
    out.orbital_energies = ea.as_mut_ptr();
    out.orbital_energies_beta = eb.as_mut_ptr();
    match unsafe { ffi::qc_scf_uhf(h.ptr, &cfg, &mut out) } {
        ffi::QC_OK => Some(UnrestrictedHartreeFockOutput { orbital_energies_alpha: ea, orbital_energies_beta: eb,
                                                          electronic_energy: out.electronic_energy, nuclear_repulsion: out.nuclear_repulsion,
                                                          iterations: out.iterations }),
        ffi::QC_NOT_CONVERGED => None,                       // uhf.rs:165-166
        ffi::QC_DIIS_SINGULAR => panic!("DIIS failed"),      // uhf.rs:95-97
        e => panic!("qchem_hip error {e}"),
    }
}

/// The convergence loop owned by Rust (the north-star wording): one FFI call per pass of rhf.rs:66-104.
pub fn restricted_hartree_fock_stepwise(system: &FlatSystem, config: &HartreeFockConfig) -> Option<RestrictedHartreeFockOutput> {
    let h = GpuSystem::new(system)?;
    let mut st = null_mut();
    if unsafe { ffi::qc_scf_begin_rhf(h.ptr, &mut st) } != ffi::QC_OK { return None; }
    let mut result = None;
    for iteration in 0..=config.max_iterations {             // inclusive, rhf.rs:66
        let (mut e, mut rms) = (0.0f64, 0.0f64);
        match unsafe { ffi::qc_scf_iterate(st, &mut e, &mut rms) } {
            ffi::QC_OK => {}
            ffi::QC_DIIS_SINGULAR => { unsafe { ffi::qc_scf_end(st) }; panic!("DIIS failed") }
            ffi::QC_EIG_NOT_CONVERGED => { unsafe { ffi::qc_scf_end(st) }; panic!("eigensolve did not converge") }
            _ => break,
        }
        if rms < config.epsilon {                            // rhf.rs:94
            let mut eps = vec![0.0f64; h.n_basis()];
            unsafe { ffi::qc_scf_orbital_energies(st, 0, eps.as_mut_ptr()) };
            result = Some(RestrictedHartreeFockOutput { orbital_energies: eps, electronic_energy: e,
                                                        nuclear_repulsion: unsafe { ffi::qc_nuclear_repulsion(h.ptr) }, iterations: iteration });
            break;
        }
    }
    unsafe { ffi::qc_scf_end(st) };
    result
}
