// Links libpixo_hip.so (built by `make -C pixo_amd/csrc`).  PIXO_HIP_LIB_DIR overrides the
// search path; the HIP runtime itself is a dependency of the shared library, not of Rust.
fn main() {
    let dir = std::env::var("PIXO_HIP_LIB_DIR").unwrap_or_else(|_| "../pixo_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pixo_hip");
    println!("cargo:rerun-if-env-changed=PIXO_HIP_LIB_DIR");
}
