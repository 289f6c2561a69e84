//! Finds libvelesdb_hip.so.  `VELESDB_HIP_LIB_DIR` names the directory that holds it (the in-tree build leaves it in
//! `velesdb_amd/`); without the variable the system linker path is used.  The library links the HIP runtime itself
//! (`libamdhip64.so` from /opt/rocm/lib) and binds RCCL with dlopen at first use, so nothing else is linked here.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=VELESDB_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
    if let Ok(dir) = env::var("VELESDB_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=velesdb_hip");
}
