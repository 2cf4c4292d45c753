// Build script of the `hip` feature (UNBUILT SOURCE: no cargo/rustc in the development image).
// LIBZKP_HIP_DIR points at the directory that holds libzkp_hip.so (this repository: libzkp_amd/lib).
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("LIBZKP_HIP_DIR").unwrap_or_else(|_| "/usr/local/lib".to_string());
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=zkp_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
        println!("cargo:rerun-if-env-changed=LIBZKP_HIP_DIR");
    }
}
