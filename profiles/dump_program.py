#!/usr/bin/env python3
"""The scene program of a workload as rr_scene_upload builds it (RR_DUMP_PROGRAM=1: one op per line on stderr).
usage: RR_DUMP_PROGRAM=1 python profiles/dump_program.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_raytracer_amd  # noqa: E402,F401
from rust_raytracer_amd import device, host  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "final"
scene = bench.WORKLOADS[workload][0]
b, flat, _ = host.build_config_scene(scene)
ctx = device.Context(0)
ctx.upload(flat)
