// Link against the in-tree engine: set GFFX_HIP_LIB_DIR to <repo>/gffx_amd/lib (holds libgffx_hip.so).
fn main() {
    if let Ok(dir) = std::env::var("GFFX_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=gffx_hip");
    println!("cargo:rerun-if-env-changed=GFFX_HIP_LIB_DIR");
}
