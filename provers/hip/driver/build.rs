// Links libraiko_hip.so.  The library itself is built by `make -C raiko_amd/csrc`
// (hipcc --offload-arch=gfx950); RAIKO_HIP_LIB_DIR points at the directory that holds it.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=RAIKO_HIP_LIB_DIR");
    let dir = env::var("RAIKO_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // default: the in-tree build next to this crate (raiko_amd/ at the repository root)
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../raiko_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=raiko_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
