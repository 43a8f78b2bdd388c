// NOT BUILT here (no Rust toolchain in the image) -- see ../README.md
// PVW_HIP_LIB_DIR = directory holding libpvw_hip.so (pvw_rs_amd/ after `python -c "import __graft_entry__ as g; g.build()"`)
fn main() {
    let dir = std::env::var("PVW_HIP_LIB_DIR").expect("set PVW_HIP_LIB_DIR to the directory of libpvw_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pvw_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=PVW_HIP_LIB_DIR");
}
