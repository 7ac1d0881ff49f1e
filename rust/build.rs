// Links libbjj_hip.so.  BJJ_HIP_LIB_DIR = directory that holds it (default: ../babyjubjub-rs_amd/csrc, where
// `python -c "import __graft_entry__ as g; g.build()"` or `make -C babyjubjub-rs_amd/csrc` leaves it).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("BJJ_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("babyjubjub-rs_amd").join("csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bjj_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BJJ_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/bjj_hip.h");
}
