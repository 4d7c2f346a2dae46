// Link against libqchem_hip.so; QCHEM_HIP_LIB_DIR = the directory `make -C qchem-rs_amd/csrc` wrote it to.
fn main() {
    let dir = std::env::var("QCHEM_HIP_LIB_DIR").expect("set QCHEM_HIP_LIB_DIR to the directory holding libqchem_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=qchem_hip");
    println!("cargo:rerun-if-env-changed=QCHEM_HIP_LIB_DIR");
}
