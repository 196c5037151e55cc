// Points the linker at libraytracer_amd.so.  RAYTRACER_AMD_LIB_DIR is the
// directory that holds the built library (raytracer_amd/ in this repository).
fn main() {
    let dir = std::env::var("RAYTRACER_AMD_LIB_DIR").unwrap_or_else(|_| "../raytracer_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=raytracer_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RAYTRACER_AMD_LIB_DIR");
}
