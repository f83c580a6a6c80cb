//! Links `libopenvm_hip.so` (the C ABI of include/openvm_hip.h, built for gfx950 by
//! `stark-backend_amd/build.sh` with hipcc). Peer of crates/cuda-backend/build.rs, which drives nvcc
//! through openvm-cuda-builder; here the library is built by its own script (hipcc cross-compiles
//! without a GPU) and only located and linked.
use std::{env, path::PathBuf, process::Command};

fn main() {
    println!("cargo:rerun-if-env-changed=OPENVM_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=OPENVM_HIP_SRC_DIR");
    let lib_dir = match env::var("OPENVM_HIP_LIB_DIR") {
        Ok(dir) => PathBuf::from(dir),
        Err(_) => {
            // build from source: OPENVM_HIP_SRC_DIR = checkout of the HIP backend (contains stark-backend_amd/)
            let src = PathBuf::from(
                env::var("OPENVM_HIP_SRC_DIR")
                    .expect("set OPENVM_HIP_LIB_DIR (prebuilt libopenvm_hip.so) or OPENVM_HIP_SRC_DIR (sources)"),
            );
            let script = src.join("stark-backend_amd").join("build.sh");
            println!("cargo:rerun-if-changed={}", src.join("stark-backend_amd/csrc").display());
            println!("cargo:rerun-if-changed={}", src.join("include/openvm_hip.h").display());
            let status = Command::new(&script).status().expect("failed to run stark-backend_amd/build.sh");
            assert!(status.success(), "stark-backend_amd/build.sh failed (is /opt/rocm/bin/hipcc installed?)");
            src.join("stark-backend_amd")
        }
    };
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=openvm_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    // the HIP runtime the library itself is linked against
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
