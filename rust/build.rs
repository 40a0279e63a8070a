// build.rs for the `ntt` crate: compiles the HIP library with hipcc and links it.
// Unbuilt in this repository's image (no cargo/rustc there); kept next to the C ABI it binds.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = PathBuf::from(env::var("NTTX_SRC").unwrap_or_else(|_| "nttx/ntt_amd/csrc".into()));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("libnttx.so");
    let status = Command::new(&hipcc)
        .args(["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared"])
        .arg(src.join("nttx.hip"))
        .arg("-o")
        .arg(&lib)
        .status()
        .expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=nttx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-changed={}", src.display());
}
