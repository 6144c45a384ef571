// Link against the two shared libraries built by `python -c "import __graft_entry__ as g; g.build()"`
// (rust-lbfgs_amd/liblbfgs_hip.so, rust-lbfgs_amd/liblbfgs_solver.so).
fn main() {
    let dir = std::env::var("LBFGS_HIP_LIB_DIR")
        .expect("set LBFGS_HIP_LIB_DIR to the directory that holds liblbfgs_hip.so and liblbfgs_solver.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lbfgs_solver");
    println!("cargo:rustc-link-lib=dylib=lbfgs_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=LBFGS_HIP_LIB_DIR");
}
