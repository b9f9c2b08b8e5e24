// Links libaha_hip.so (built by `python -m aha_amd.build`: hipcc --offload-arch=gfx950, in-tree at aha_amd/csrc/).
// AHA_HIP_LIB_DIR overrides the directory.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("AHA_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../aha_amd/csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=aha_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=AHA_HIP_LIB_DIR");
}
