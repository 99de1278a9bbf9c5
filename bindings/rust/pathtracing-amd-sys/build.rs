// Tells rustc where libpathtracing_amd.so lives: PATHTRACING_AMD_LIB_DIR, or the
// in-tree build directory of this repository (pathtracing_amd/).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var_os("PATHTRACING_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("../../../pathtracing_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=pathtracing_amd");
    println!("cargo:rerun-if-env-changed=PATHTRACING_AMD_LIB_DIR");
}
