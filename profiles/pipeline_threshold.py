"""Path kernel against wavefront pipeline on all-BVH scenes of growing size (height field + spheres): the measurement behind the
size threshold of rr_scene_info.wavefront.  Run on the GPU box from the repository root: python profiles/pipeline_threshold.py"""
import sys, time
sys.path.insert(0, '.')
import rust_raytracer_amd  # noqa
from rust_raytracer_amd import abi, device, host
ctx = device.Context(0)
for nt, ns in [(5000, 500), (20000, 2000), (60000, 6000), (200000, 20000), (500000, 50000)]:
    b, flat, _ = host.build_config_scene("mesh", n_triangles=nt, n_spheres=ns)
    info = ctx.upload(flat)
    out = []
    for flags in (abi.FLAG_PATHS, abi.FLAG_WAVEFRONT):
        p = device.make_params(1280, 720, 16, n_slices=8, max_depth=50, seed=1, flags=flags)
        ctx.render(p, slice_sums=False)
        ctx.render(p, slice_sums=False)
        out.append(ctx.last_kernel_ms())
    print(nt, ns, "bytes", info["device_bytes"], "auto wavefront", info["wavefront"], "paths ms %.2f pipeline ms %.2f" % tuple(out), flush=True)
