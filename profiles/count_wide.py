import sys, json
sys.path.insert(0,'/root/repo')
import rust_raytracer_amd
from rust_raytracer_amd import abi, device, host
ctx = device.Context(0)
for name, W, H, spp, depth in (("mesh", 640, 360, 8, 50), ("final", 200, 200, 32, 50), ("suzanne", 200, 200, 32, 100), ("spheres", 200, 112, 32, 50)):
    b, flat, _ = host.build_config_scene(name)
    info = ctx.upload(flat)
    for fl in (0, abi.FLAG_NARROW_NODES):
        p = device.make_params(W, H, spp, n_slices=8, max_depth=depth, seed=1, flags=fl | abi.FLAG_COUNT_PRODUCTION)
        r = ctx.render(p, slice_sums=False)
        c = r["counters"]; rays = c["rays"]
        print(name, "wide" if fl == 0 else "narrow", info["wide_nodes"], {k: round(v / rays, 2) for k, v in c.items() if k not in ("samples", "rays")},
              "lanes", round(c["machine_lane_steps"] / max(1, c["machine_wave_steps"]), 2), flush=True)
