#!/usr/bin/env python3
"""How evenly the two ways of sharing ONE image over N GPUs load them, measured on one GPU: the kernel time of each of the N
bands of rows (contiguous rows, all slices) against the time of one of N slice groups (whole image, K/N slices).  The slowest
share is the time of the N-GPU job.  usage: python profiles/band_balance.py [workload] [spp] [N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_raytracer_amd  # noqa: E402,F401
from rust_raytracer_amd import abi, device, host, multi  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "final"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scene, W, H, _, depth = bench.WORKLOADS[workload]
b, flat, _ = host.build_config_scene(scene)
ctx = device.Context(0)
ctx.upload(flat)
dev = torch.device("cuda", 0)
job = multi.job_params(W, H, spp, 8, depth, 1)
out = {"workload": workload, "spp": spp, "n": N}
for part in ("rows", "slices"):
    ms = []
    for r in range(N):
        p = multi.rank_share(job, r, N, part)
        k, rows = p.slice_end - p.slice_begin, p.rows
        sums = torch.empty((max(1, k), rows, W, 3), dtype=torch.float32, device=dev)
        for _ in range(2):
            ctx.render_device(p, sums.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        ms.append(round(ctx.last_kernel_ms(), 3))
    out[part] = {"ms_per_share": ms, "max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4), "job_ms": max(ms)}
print(json.dumps(out))
