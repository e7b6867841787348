#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload with the scene already resident in HBM.  The default workload is the
configuration the throughput target is quoted on: final_scene (src/main.rs:466-541, light repaired as cornell_box
has it, one moving sphere) at 800x800, 10 000 samples per pixel, depth 50, K = 8 sample slices.  With N > 1
(launched by torch.distributed.run, one rank per GPU) the ranks share ONE render_multi job:

  --scaling weak   (default) a job of N x the samples in N x the slices: the work per GPU is fixed;
  --scaling strong the job of one GPU, whatever N: the time per image falls with N;
  --partition slices|rows|both   how the job is cut: sample slices (summed with one reduce), bands of image
                   rows (gathered), or both (a reduce per band, a gather over bands).

The combination of the ranks' images (RCCL) and the u8 cast are inside the timed region.  Rank 0 prints one JSON
line.  `value` is whole-job Msamples/s; `roofline` prices the path-tracing kernels by what they did (tests counted by
an instrumented pass of the production traversal); `cpu_baseline` is the CPU oracle (oracle/librr_oracle.so, a C++
restatement of the reference: the Rust reference itself cannot be built here) timed on this box's host cores on a
bounded sample of the same workload.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak
VALU_PEAK_TFLOPS = 157.3    # MI355X peak vector f32, counting a fused multiply-add as two operations

WORKLOADS = {
    # name: (scene, width, height, spp, depth): the benchmark configurations
    "final": ("final", 800, 800, 10000, 50),
    "cornell": ("cornell", 600, 600, 1000, 200),
    "suzanne": ("suzanne", 800, 800, 1000, 100),
    "spheres": ("spheres", 400, 225, 100, 50),
    "mesh": ("mesh", 3840, 2160, 4096, 50),   # 1M-triangle height field + 100k spheres; use --spp for a short run
}
COUNTER_NAMES = ["samples", "rays", "box_tests", "prim_tests", "inst_visits", "sphere_tests", "rect_tests", "tri_tests",
                 "machine_lane_steps", "machine_wave_steps"]


def host_threads():
    n = len(os.sched_getaffinity(0))
    env = os.environ.get("OMP_NUM_THREADS")
    cap = int(env) if env and env.isdigit() else 16
    return max(1, min(n, cap))


def bytes_per_ray(c):
    """Record bytes per ray for the tests counted in c: 32 B per box test, 16 B per sphere, 24 B per rect, 48 B per
    triangle, 16 B per instance or medium visit."""
    rays = max(1, c["rays"])
    return (32.0 * c["box_tests"] + 16.0 * c["sphere_tests"] + 24.0 * c["rect_tests"] + 48.0 * c["tri_tests"] +
            16.0 * c["inst_visits"]) / rays


def flops_per_ray(c):
    """f32 operations per ray for the tests counted in c: 20 per box test, 30 per sphere, 15 per rect, 60 per
    triangle, plus 150 for shading and sampling."""
    rays = max(1, c["rays"])
    return (20.0 * c["box_tests"] + 30.0 * c["sphere_tests"] + 15.0 * c["rect_tests"] + 60.0 * c["tri_tests"]) / rays + 150.0


def cpu_info():
    model, mhz = "unknown", []
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("cpu MHz"):
                mhz.append(float(line.split(":", 1)[1]))
    except OSError:
        pass
    return model, (round(statistics.median(mhz)) if mhz else None)


def cpu_baseline(flat, width, height, depth, seed):
    """Time the CPU oracle on this host: one full-image sample slice per thread (render_multi's own decomposition,
    render.rs:128-151), the process confined to as many cores as it runs threads, median of three runs of a few
    seconds each on a reduced image of the same scene and camera (throughput per sample does not depend on the
    image size: the pixels of the small image are a regular subsample of the large one's)."""
    import oracle_lib as O
    threads = host_threads()
    allowed = sorted(os.sched_getaffinity(0))
    before = set(allowed)
    try:
        os.sched_setaffinity(0, set(allowed[:threads]))  # the oracle's worker threads inherit the mask
    except OSError:
        pass
    w = min(width, 200)
    h = max(1, min(height, round(height * w / width)))

    def timed(spt):
        p = O.make_params(w, h, threads * spt, n_slices=threads, max_depth=depth, seed=seed)
        t0 = time.perf_counter()
        r = O.render(flat, p, n_threads=threads, counters=True)
        return time.perf_counter() - t0, r

    dt, r = timed(1)  # calibration (also warms caches and clocks up)
    spt = int(max(1, min(4096, round(5.0 / max(dt, 1e-3)))))
    runs = []
    for _ in range(3):
        dt, r = timed(spt)
        runs.append((w * h * threads * spt / dt / 1e6, r["counters"]["rays"] / dt / 1e6, dt))
    try:
        os.sched_setaffinity(0, before)
    except OSError:
        pass
    runs.sort()
    med = runs[1]
    model, mhz = cpu_info()
    return {
        "value": round(med[0], 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{w}x{h} (the {width}x{height} camera), {threads * spt} spp in {threads} slices, depth {depth}: "
                  f"{w * h * threads * spt} samples per run, median of 3 runs ({', '.join(f'{x[2]:.1f} s' for x in runs)})",
        "runs_msamples_per_s": [round(x[0], 4) for x in runs],
        "mrays_per_s": round(med[1], 3),
        "cpu_model": model, "cpu_mhz_median": mhz, "confined_to_cpus": allowed[:threads],
        "note": "C++ restatement of the reference (no per-hit Material clones, Box<PDF> allocations or atlas mutex), "
                "so faster than the Rust binary would be: GPU/CPU ratios against it are conservative; process confined to "
                "`cores` CPUs (sched_setaffinity), median of three runs",
    }


def latest_profile(pattern):
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            return json.load(open(path)), os.path.basename(path)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="final", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (of the job under strong scaling, per GPU under weak)")
    ap.add_argument("--slices", type=int, default=8, help="sample slices K, the reference's n_threads (of the job / per GPU)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--partition", default="slices", choices=["slices", "rows", "both"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--flags", type=int, default=0, help="RR_FLAG_* bits for the timed renders (kernel organisation)")
    ap.add_argument("--semantics", default="reference", choices=["reference", "intended"],
                    help="intended: the labelled alternative semantics (RR_SEMANTICS_INTENDED), never the default")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented counting passes")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import abi, device, host, multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    scene_name, W, H, spp, depth = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    K = args.slices
    builder, flat, _ = host.build_config_scene(scene_name, seed=args.seed)
    if args.semantics == "intended":
        flat.semantics = abi.SEMANTICS_INTENDED
    ctx = device.Context(local_rank)
    info = ctx.upload(flat)

    job = multi.job_params(W, H, spp, K, depth, args.seed, world, args.scaling, flags=args.flags)
    p = multi.rank_share(job, rank, world, args.partition)
    comb = multi.Combiner(dist, world, args.partition, H)
    n_sl, rows = p.slice_end - p.slice_begin, p.rows
    spt = job.samples_per_pixel // job.n_slices
    dev = torch.device("cuda", local_rank)
    slice_sums = torch.empty((max(1, n_sl), rows, W, 3), dtype=torch.float32, device=dev)
    resolved = torch.empty((rows, W, 3), dtype=torch.float32, device=dev)
    linear = torch.empty((rows, W, 3), dtype=torch.float32, device=dev)
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    counters = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(ev=None):
        if ev:
            ev[0].record()
        ctx.render_device(p, slice_sums.data_ptr(), resolved.data_ptr(), linear.data_ptr(), counters.data_ptr(), stream)
        if ev:
            ev[1].record()
        whole = comb.combine((resolved, linear))
        if rank == 0:
            ctx.cast_u8_device(whole[0].data_ptr(), rgb8.data_ptr(), whole[0].numel(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    counters.zero_()
    torch.cuda.synchronize()
    # HIP events on the stream the kernels are launched on (torch's current stream is the one handed to the library;
    # what the library runs on its second stream is ordered inside the bracket by its own events)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = counters.clone()
        dist.all_reduce(tot)
        counts_all = tot.cpu().numpy()
    else:
        counts_all = counters.cpu().numpy()
    counts_rank = counters.cpu().numpy()

    if rank == 0:
        samples_per_step = W * H * spt * job.n_slices
        total_samples = int(counts_all[0])
        total_rays = int(counts_all[1])
        assert total_samples == samples_per_step * args.steps, (total_samples, samples_per_step, args.steps)
        value = total_samples / elapsed / 1e6
        rays_per_launch = int(counts_rank[1]) / args.steps
        samples_per_launch = int(counts_rank[0]) / args.steps
        k_ms = float(np.mean(kernel_ms))
        if args.flags & abi.FLAG_PERSISTENT:
            kernel_name = "k_render (item-granular persistent kernel)"
        elif (args.flags & abi.FLAG_WAVEFRONT) or (info["wavefront"] and not (args.flags & abi.FLAG_PATHS)):
            kernel_name = "wavefront pipeline: k_wf_init + k_wf_traverse + k_wf_shade (camera rays) + k_wf_finish + k_wf_accumulate"
        else:
            kernel_name = "k_wf_paths (persistent kernel at path granularity) + k_wf_accumulate"

        # What the kernels did per ray, from instrumented passes outside the timed region on a reduced image (per-ray
        # averages do not depend on the image size): `production` counts the tests of the traversal the timed kernels run
        # (redundant visits skipped, subtrees culled); `reference_order` counts every visit the reference's algorithm makes.
        per_ray = {}
        if not args.no_count:
            scale = min(1.0, (2.0e6 / (W * H * K * 4.0)) ** 0.5)
            Wc, Hc = max(16, int(W * scale)), max(16, int(H * scale))
            for label, flag in (("production", abi.FLAG_COUNT_PRODUCTION), ("reference_order", abi.FLAG_COUNTERS)):
                pc = device.make_params(Wc, Hc, K * 4, n_slices=K, max_depth=depth, seed=args.seed,
                                        flags=flag | (args.flags & (abi.FLAG_PERSISTENT | abi.FLAG_WAVEFRONT | abi.FLAG_PATHS)))
                c2 = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)
                cs = torch.empty((K, Hc, Wc, 3), dtype=torch.float32, device=dev)
                ctx.render_device(pc, cs.data_ptr(), None, None, c2.data_ptr(), stream)
                torch.cuda.synchronize()
                per_ray[label] = dict(zip(COUNTER_NAMES, [int(x) for x in c2.cpu().numpy()]))
            print(f"# counting passes: {Wc}x{Hc}, {K * 4} spp", file=sys.stderr, flush=True)
        prod = per_ray.get("production")
        roofline = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                    "kernel": kernel_name, "kernel_ms": round(k_ms, 3), "rays_per_launch": int(rays_per_launch)}
        if prod:
            fpr, bpr = flops_per_ray(prod), bytes_per_ray(prod)
            achieved = fpr * rays_per_launch / (k_ms * 1e-3) / 1e12
            ref = per_ray["reference_order"]
            rays_c = max(1, prod["rays"])
            roofline.update({
                "achieved": round(achieved, 3), "frac": round(achieved / VALU_PEAK_TFLOPS, 5),
                "peak_without_fma": VALU_PEAK_TFLOPS / 2, "frac_of_peak_without_fma": round(2 * achieved / VALU_PEAK_TFLOPS, 5),
                "flops_per_ray": round(fpr, 1),
                "per_ray_production": {k: round(prod[k] / rays_c, 3) for k in COUNTER_NAMES[2:8]},
                "per_ray_reference_order": {k: round(ref[k] / max(1, ref["rays"]), 3) for k in COUNTER_NAMES[2:8]},
                "machine_lanes_at_work": (round(prod["machine_lane_steps"] / max(1, prod["machine_wave_steps"]), 2)
                                          if prod["machine_wave_steps"] else None),
                "record_bytes_per_ray": round(bpr, 1),
                "record_gbs": round((bpr * rays_per_launch + 12.0 * samples_per_launch) / (k_ms * 1e-3) / 1e9, 1),
                "note": "bound = vector ALU issue: the scene records are cache-resident (scalar cache / L2; `record_gbs` is the "
                        "rate at which the kernels consume record bytes, served by caches, not HBM: see traffic and hbm_frac). "
                        "achieved = flops_per_ray (20/box test, 30/sphere, 15/rect, 60/triangle, +150 shading; tests counted on the "
                        "production traversal) x rays per launch / kernel time.  The parity contract forbids fused multiply-add "
                        "(-ffp-contract=off), so the attainable peak is peak_without_fma."})
        # HBM bytes per launch and VALU issue figures from the rocprofv3 PMC passes of the same workload kept under profiles/
        tr, tr_src = latest_profile("r*_pmc_traffic.json")
        if tr and tr.get(args.workload, {}).get("hbm_bytes_per_sample") is not None:
            traffic = tr[args.workload]["hbm_bytes_per_sample"] * samples_per_launch
            roofline["traffic"] = round(traffic)
            roofline["hbm_frac"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roofline["traffic_source"] = tr_src
        summ, summ_src = latest_profile(f"r*_{args.workload}_summary.json")
        if summ:
            dd = summ.get("derived", {})
            roofline["measured"] = {k: dd.get(k) for k in ("kernel", "valu_issue_slot_frac", "active_lanes_per_valu_inst",
                                                          "valu_insts_per_wave_ray", "salu_insts_per_wave_ray", "waves_per_cu",
                                                          "lane_slot_frac") if k in dd}
            roofline["measured"]["source"] = summ_src
        out = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {job.samples_per_pixel} spp in {job.n_slices} slices, depth {depth}"
                                   + (f" ({spp} spp in {K} slices per GPU)" if args.scaling == "weak" and world > 1 else ""),
                       "scene": scene_name, "width": W, "height": H, "spp": job.samples_per_pixel, "slices": job.n_slices,
                       "max_depth": depth, "parallelism": f"{args.partition} x{world} ({comb.bands} bands x {comb.groups} slice groups)",
                       "seed": args.seed, "flags": args.flags, "semantics": args.semantics, "stack_words": info["stack_words"], "scene_bytes": info["device_bytes"]},
            "mrays_per_s": round(total_rays / elapsed / 1e6, 3),
            "rays_per_sample": round(total_rays / max(1, total_samples), 4),
            "kernel_ms_per_step": [round(x, 3) for x in kernel_ms],
            "roofline": roofline,
            "reference_published": {"msamples_per_s": 0.67, "source": "src/notes.txt:1-5 (Cornell box 600x600, 200 spp, depth 200, unknown CPU)"},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(flat, W, H, depth, args.seed)
            out["speedup_vs_cpu_baseline"] = round(value / max(1e-9, out["cpu_baseline"]["value"]), 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
