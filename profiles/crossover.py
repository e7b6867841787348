#!/usr/bin/env python3
"""Path kernel against wavefront pipeline on height fields with spheres of growing size (1280x720, 16 spp): where the scene
stops being cache-resident the pipeline takes over (rr_scene_upload's threshold).  usage: python profiles/crossover.py
[RR_HIP_LIB=<variant> to measure another build]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_raytracer_amd  # noqa: E402,F401
from rust_raytracer_amd import abi, device, host  # noqa: E402

W, H, SPP, SLICES = 1280, 720, 16, 8
ctx = device.Context(0)
sums = torch.empty((SLICES, H, W, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for n_tri in (3_000, 12_000, 20_000, 26_000, 33_000, 120_000, 300_000, 600_000):
    b, flat, _ = host.build_config_scene("mesh", n_triangles=n_tri, n_spheres=n_tri // 10)
    info = ctx.upload(flat)
    row = {"triangles": n_tri, "spheres": n_tri // 10, "scene_mb": round(info["device_bytes"] / 1e6, 2), "pipeline_is_default": bool(info["wavefront"])}
    for name, flags in (("path_kernel_ms", abi.FLAG_PATHS), ("pipeline_ms", abi.FLAG_WAVEFRONT)):
        p = device.make_params(W, H, SPP, n_slices=SLICES, max_depth=50, seed=1, flags=flags)
        best = None
        for _ in range(3):
            ctx.render_device(p, sums.data_ptr(), None, None, None, stream)
            torch.cuda.synchronize()
            ms = ctx.last_kernel_ms()
            best = ms if best is None else min(best, ms)
        row[name] = round(best, 2)
    print(json.dumps(row), flush=True)
