// build.rs for the `ntt` crate: builds the HIP library with its own Makefile (hipcc, one object
// per arithmetic policy) and links it.  Unbuilt in this repository's image (no cargo/rustc there);
// kept next to the C ABI it binds.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    // checkout of the nttx sources (include/, ntt_amd/csrc/, ntt_amd/host/)
    let root = PathBuf::from(env::var("NTTX_SRC").unwrap_or_else(|_| "nttx".into()));
    let csrc = root.join("ntt_amd/csrc");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let jobs = env::var("NUM_JOBS").unwrap_or_else(|_| "8".into());
    let status = Command::new("make")
        .arg(format!("-j{jobs}"))
        .arg("-C")
        .arg(&csrc)
        .arg(format!("HIPCC={hipcc}"))
        .arg(format!("OUT={}", out.join("libnttx.so").display()))
        .arg(format!("OBJDIR={}", out.join("obj").display()))
        .status()
        .expect("failed to run make");
    assert!(status.success(), "building libnttx.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=nttx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", root.join("include/nttx.h").display());
}
