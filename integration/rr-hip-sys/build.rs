// Builds librr_hip.so from the one HIP translation unit and links it.
// RR_HIP_SRC names the checkout of the backend (the directory that holds include/ and rust-raytracer_amd/);
// HIPCC overrides the compiler (default /opt/rocm/bin/hipcc).  -ffp-contract=off is part of the arithmetic
// contract: one IEEE f32 operation per source operation, as the Rust code computes.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("RR_HIP_SRC").expect("set RR_HIP_SRC to the backend checkout"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = root.join("rust-raytracer_amd/csrc/rr_hip.hip");
    let so = out.join("librr_hip.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"])
        // rr_device.h includes "rr_hip.h"; the library also embeds the text of its device headers (for rr_scene_specialize),
        // which the assembler looks for along the same paths
        .arg("-I")
        .arg(root.join("rust-raytracer_amd/csrc"))
        .arg("-I")
        .arg(root.join("include"))
        .arg(&src)
        .arg("-o")
        .arg(&so)
        .status()
        .unwrap_or_else(|e| panic!("{hipcc}: {e}"))
        .success();
    assert!(ok, "hipcc failed on {}", src.display());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rr_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    for f in ["rust-raytracer_amd/csrc/rr_hip.hip", "rust-raytracer_amd/csrc/rr_device.h", "rust-raytracer_amd/csrc/rr_wavefront.h", "include/rr_hip.h"] {
        println!("cargo:rerun-if-changed={}", root.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=RR_HIP_SRC");
    println!("cargo:rerun-if-env-changed=HIPCC");
}
