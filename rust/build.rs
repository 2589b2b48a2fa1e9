fn main() {
    // libcntt_hip.so lives in <repo>/concrete-ntt_amd/
    let dir = std::env::var("CNTT_HIP_LIB_DIR").unwrap_or_else(|_| "../concrete-ntt_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=cntt_hip");
}
