// Links libbls381_hip.so (built by `make -C plonky2_bls12_381_pairing_amd/csrc`, hipcc --offload-arch=gfx950).
// BLS381_HIP_LIB_DIR names the directory that holds it; the default is the in-tree build location.
fn main() {
    let dir = std::env::var("BLS381_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../plonky2_bls12_381_pairing_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bls381_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BLS381_HIP_LIB_DIR");
}
