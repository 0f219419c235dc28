// Points the linker at the in-tree liborcd.so (built by `python -m orc_format_amd.build`).
fn main() {
    let dir = std::env::var("ORCD_LIB_DIR").unwrap_or_else(|_| "../orc_format_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=orcd");
}
