#!/usr/bin/env python3
"""Where a turn of the path kernel spends its wave clocks, by section (a build with -DRR_SECTION_TIMERS: timing only, slower
than production because every mark drains the scalar-load counter).  usage: RR_HIP_LIB=<timing build> python profiles/sections.py [workload] [spp]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_raytracer_amd  # noqa: E402,F401
from rust_raytracer_amd import abi, device, host, multi  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "final"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene, W, H, _, depth = bench.WORKLOADS[workload]
b, flat, _ = host.build_config_scene(scene)
ctx = device.Context(0)
ctx.upload(flat)
lib = C.CDLL(abi.HIP_LIB)  # (the same handle the context uses: dlopen returns it again)
fn = lib.rr_debug_section_cycles
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
job = multi.job_params(W, H, spp, 8, depth, 1)
p = multi.rank_share(job, 0, 1, "slices")
sums = torch.empty((8, H, W, 3), dtype=torch.float32, device="cuda:0")
ctx.render_device(p, sums.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert fn(buf) == 0
ctx.render_device(p, sums.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ms = ctx.last_kernel_ms()
assert fn(buf) == 0
v = list(buf)
names = ["start paths", "ops outside subtrees", "machine loop, first subtree", "machine loop, later subtrees", "subtree entry + park settle", "shade", "store", "loop top"]
tot = sum(v[:8])
turns, lanes = v[10], v[11]
out = {"workload": workload, "spp": spp, "kernel_ms": round(ms, 2), "turns": turns, "lanes_running_per_turn": round(lanes / max(1, turns), 2),
       "cycles_per_turn": round(tot / max(1, turns), 1),
       "share": {n: round(v[i] / tot, 4) for i, n in enumerate(names)},
       "cycles_per_turn_by_section": {n: round(v[i] / max(1, turns), 1) for i, n in enumerate(names)},
       "machine_steps_per_turn": [round(v[8] / max(1, turns), 2), round(v[9] / max(1, turns), 2)],
       "cycles_per_machine_step": [round(v[2] / max(1, v[8]), 1), round(v[3] / max(1, v[9]), 1)],
       "lanes_per_machine_step": [round(v[14] / max(1, v[8]), 2), round(v[15] / max(1, v[9]), 2)],
       "lanes_entering_per_turn": [round(v[12] / max(1, turns), 2), round(v[13] / max(1, turns), 2)]}
print(json.dumps(out, indent=1))
