// Links librusty_compression_amd.so (built by `make -C rusty_compression_amd/csrc`).
fn main() {
    let dir = std::env::var("RUSTY_COMPRESSION_AMD_LIB_DIR").unwrap_or_else(|_| "../../rusty_compression_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rusty_compression_amd");
}
