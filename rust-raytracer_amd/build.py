"""In-tree builds: librr_hip.so (hipcc, gfx950), librr_host.so (g++), oracle/librr_oracle.so (make), examples/render_scene (gcc).

hipcc cross-compiles without a GPU, so all three build on a CPU-only box.  A library is
rebuilt when it is older than any of its sources.
"""
import os
import subprocess
import tempfile

from . import abi

CSRC = os.path.join(abi.PKG_DIR, "csrc")
HOST = os.path.join(abi.PKG_DIR, "host")
INCLUDE = os.path.join(abi.REPO_DIR, "include")

# -ffp-contract=off: the path's arithmetic contract is one IEEE f32 operation per source
# operation, as in the reference's Rust (no fused multiply-add), so that ids, t and normals
# agree with the CPU oracle bit for bit.
# -fno-slp-vectorize: left to itself the compiler pairs independent f32 operations of the intersectors into packed
# instructions (v_pk_mul_f32, v_pk_add_f32), whose operands must sit in adjacent registers: in these kernels that cost about 300
# register moves and the registers to hold the pairs (the path kernel spilled 24 of them; without the pairing none), for
# instructions that issue at the rate of the two they replace.  Measured: final_scene +8.5 %, Suzanne +9.7 %, random spheres
# +11 %, Cornell box +4.4 %, the 1.1 M-primitive mesh +6 %; same bits.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-Wall"]
# rr_paths_minreg.hip, the library's second translation unit (the builds of the path kernel for scenes with one simple BVH
# subtree): compiled with the instruction scheduler that minimises register use, which fits those builds into the 96
# registers of 5 waves per SIMD with fewer spills (Suzanne +5.7 %, random spheres +2.3 %; on the other builds -0.6 ... 0 %,
# so they keep the default).  Same source, same bits.
MINREG_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"]
GXX_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)


def _headers():
    return [os.path.join(INCLUDE, "rr_hip.h"), os.path.join(INCLUDE, "rr_host.h")]


def source_digest():
    """sha256 over the device sources (csrc/*, sorted by name) and the hipcc flags: names the kernels a measurement was
    taken on, so that counter summaries kept beside the code can be told from stale ones."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    h.update(" ".join(MINREG_FLAGS).encode())
    return h.hexdigest()


def build_hip(force=False):
    units = [("rr_hip.hip", []), ("rr_paths_minreg.hip", MINREG_FLAGS)]
    src = [os.path.join(CSRC, u) for u, _ in units]
    deps = src + [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + _headers()
    if force or _stale(abi.HIP_LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        # (-I: rr_device.h includes "rr_hip.h", and the library embeds the text of its device headers, which the assembler
        # looks for along the same paths; kept out of HIPCC_FLAGS so that source_digest() does not depend on where the tree lies)
        compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-I", CSRC, "-I", INCLUDE, "-c"]
        objs = []
        with tempfile.TemporaryDirectory() as tmp:
            procs = []
            for (unit, extra), path in zip(units, src):
                obj = os.path.join(tmp, unit + ".o")
                objs.append(obj)
                procs.append((subprocess.Popen([hipcc] + compile_flags + extra + ["-o", obj, path], stdout=subprocess.PIPE,
                                               stderr=subprocess.STDOUT, text=True), unit))
            for proc, unit in procs:  # (the units compile side by side)
                out, _ = proc.communicate()
                if proc.returncode != 0:
                    raise RuntimeError("build failed: " + unit + "\n" + out)
            _run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", abi.HIP_LIB] + objs)
    return abi.HIP_LIB


def build_host(force=False):
    src = [os.path.join(HOST, f) for f in ("scene.cpp", "flatten.cpp", "scenes.cpp", "host_capi.cpp")]
    deps = src + [os.path.join(HOST, "scene.hpp")] + _headers()
    if force or _stale(abi.HOST_LIB, deps):
        _run([os.environ.get("CXX", "g++")] + GXX_FLAGS + ["-o", abi.HOST_LIB] + src)
    return abi.HOST_LIB


def build_oracle(force=False):
    d = os.path.join(abi.REPO_DIR, "oracle")
    if force:
        _run(["make", "-C", d, "clean"])
    _run(["make", "-C", d])
    return os.path.join(d, "librr_oracle.so")


def build_example(force=False):
    """examples/render_scene: the two C ABIs used from plain C (gcc -std=c99), linked against the in-tree libraries."""
    d = os.path.join(abi.REPO_DIR, "examples")
    src = os.path.join(d, "render_scene.c")
    exe = os.path.join(d, "render_scene")
    if force or _stale(exe, [src, abi.HIP_LIB, abi.HOST_LIB] + _headers()):
        _run([os.environ.get("CC", "gcc"), "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-I", INCLUDE, src, "-L", abi.PKG_DIR,
              "-lrr_hip", "-lrr_host", "-Wl,-rpath,$ORIGIN/../" + os.path.basename(abi.PKG_DIR), "-Wl,-rpath-link," + "/opt/rocm/lib", "-o", exe])
    return exe


def build_all(force=False):
    return [build_hip(force), build_host(force), build_oracle(force), build_example(force)]
