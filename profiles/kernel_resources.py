#!/usr/bin/env python3
"""Per-kernel register use of the device library as the compiler reports it (-Rpass-analysis=kernel-resource-usage):
vector and scalar registers, spills, scratch bytes per lane and the occupancy they allow.
usage: python profiles/kernel_resources.py [substring of the kernel names to show ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_raytracer_amd  # noqa: E402,F401
from rust_raytracer_amd import build  # noqa: E402

remarks = ""
for unit, extra in (("rr_hip.hip", []), ("rr_paths_minreg.hip", build.MINREG_FLAGS)):  # the library's two translation units
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + [f for f in build.HIPCC_FLAGS if f != "-shared"] + extra +
                       ["-I", build.CSRC, "-I", build.INCLUDE, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/rr_resources.o",
                        os.path.join(build.CSRC, unit)], capture_output=True, text=True)
    remarks += r.stderr
want = sys.argv[1:] or ["k_wf_paths", "k_wf_traverse", "k_wf_finish", "k_render", "k_wf_shade"]
for b in re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]:
    name = b.split("\n")[0].split()[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    if not any(k in dem for k in want):
        continue

    def g(key):
        m = re.search(key + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-78s VGPR %4s spill %3s | SGPR %4s spill %3s | scratch %4s B/lane | %s waves/SIMD" % (
        dem, g("VGPRs"), g("VGPRs Spill"), g("SGPRs"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
