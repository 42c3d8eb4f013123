// Links librtcore.so.  RTCORE_LIB_DIR names the directory that holds it (the
// library is built in-tree by `make -C rust-raytracer_amd/csrc`); at run time
// it and ROCm's libamdhip64.so must be on the loader path.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("RTCORE_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../rust-raytracer_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rtcore");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RTCORE_LIB_DIR");
}
