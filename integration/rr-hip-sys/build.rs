// Builds librr_hip.so from its two HIP translation units and links it.
// RR_HIP_SRC names the checkout of the backend (the directory that holds include/ and rust-raytracer_amd/);
// HIPCC overrides the compiler (default /opt/rocm/bin/hipcc).  -ffp-contract=off is part of the arithmetic
// contract: one IEEE f32 operation per source operation, as the Rust code computes.  The second unit
// (rr_paths_minreg.hip: some builds of the path kernel) is compiled with the instruction scheduler that
// minimises register use; same source, same results.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("RR_HIP_SRC").expect("set RR_HIP_SRC to the backend checkout"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let so = out.join("librr_hip.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let units: [(&str, &[&str]); 2] =
        [("rr_hip.hip", &[]), ("rr_paths_minreg.hip", &["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"])];
    let mut objects = Vec::new();
    for (unit, extra) in units {
        let src = root.join("rust-raytracer_amd/csrc").join(unit);
        let obj = out.join(format!("{unit}.o"));
        let ok = Command::new(&hipcc)
            .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-c"])
            .args(extra)
            // rr_device.h includes "rr_hip.h"; the library also embeds the text of its device headers (for rr_scene_specialize),
            // which the assembler looks for along the same paths
            .arg("-I")
            .arg(root.join("rust-raytracer_amd/csrc"))
            .arg("-I")
            .arg(root.join("include"))
            .arg(&src)
            .arg("-o")
            .arg(&obj)
            .status()
            .unwrap_or_else(|e| panic!("{hipcc}: {e}"))
            .success();
        assert!(ok, "hipcc failed on {}", src.display());
        objects.push(obj);
    }
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-fPIC", "-shared"])
        .args(&objects)
        .arg("-o")
        .arg(&so)
        .status()
        .unwrap_or_else(|e| panic!("{hipcc}: {e}"))
        .success();
    assert!(ok, "hipcc failed to link {}", so.display());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rr_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    for f in ["rust-raytracer_amd/csrc/rr_hip.hip", "rust-raytracer_amd/csrc/rr_paths_minreg.hip", "rust-raytracer_amd/csrc/rr_device.h", "rust-raytracer_amd/csrc/rr_wavefront.h", "include/rr_hip.h"] {
        println!("cargo:rerun-if-changed={}", root.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=RR_HIP_SRC");
    println!("cargo:rerun-if-env-changed=HIPCC");
}
