//! Links libbellman_hip.so (built by `make -C bellman_amd/csrc`, see the repository root).
//! BELLMAN_HIP_LIB_DIR = directory holding libbellman_hip.so (default: ../../bellman_amd/lib relative to this
//! crate, i.e. the in-tree build).  The library itself needs the ROCm runtime (libamdhip64) at run time.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=BELLMAN_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
    let dir = env::var_os("BELLMAN_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR")).join("../../bellman_amd/lib")
    });
    let lib = dir.join("libbellman_hip.so");
    if !lib.exists() {
        panic!(
            "{} not found: build it with `make -C bellman_amd/csrc` or point BELLMAN_HIP_LIB_DIR at it \
             (there is no CPU fallback inside the library; bellman's own CPU path stays available when \
             the `hip` feature is off or no device is present)",
            lib.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bellman_hip");
    // so that test binaries find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
