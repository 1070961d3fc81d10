// Links libpsf_mi355x.so.  PSF_MI355X_LIB_DIR = directory holding the library (default: ../tools_amd/lib, where
// `python -c "import __graft_entry__ as g; g.build()"` or `make -C tools_amd/csrc` puts it).
fn main() {
    let dir = std::env::var("PSF_MI355X_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{manifest}/../tools_amd/lib")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=psf_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=PSF_MI355X_LIB_DIR");
}
