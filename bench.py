#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload with the scene already resident in HBM.  The default workload is the
configuration the throughput target is quoted on: final_scene (src/main.rs:466-541, light repaired as cornell_box
has it, one moving sphere) at 800x800, 10 000 samples per pixel, depth 50, K = 8 sample slices.  With N > 1 the
N GPUs share ONE render_multi job, one rank process per GPU over RCCL: either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), or `python bench.py --gpus N` starts them
itself as a torch.distributed.run child before it has touched the GPU.  It never measures fewer GPUs than it was asked
for: if fewer than N are visible, or the world it finds itself in is not N, it exits non-zero.

  --scaling strong (default) the job of one GPU, whatever N (the stated job: ONE 10 000-spp image over the node): the
                   time per image falls with N;
  --scaling weak   a job of N x the samples in N x the slices: the work per GPU is fixed.  With N > 1 and the
                   default scaling the weak job is measured as well and reported in the field `weak`;
  --partition slices|rows|both   how the job is cut: sample slices (summed with one reduce), bands of image
                   rows (gathered), or both (a reduce per band, a gather over bands);
  --inprocess [--devices 0,1,..]   one process drives all N devices through the library's rr_multi_* entry points
                   (peer copies or RCCL inside the library) instead of one rank per GPU.

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
GATHER_PEAK_GBS = 8600.0    # MI355X Infinity Cache, random 64-byte rows (the scene of the mesh workload is resident in it)

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


def cgroup_cpu_quota():
    """CPUs this process may use at once under its cgroup's CFS quota (None: no quota)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return None


def host_threads():
    """(threads to run, cores in the affinity mask, cgroup quota): every core this process may use, as the reference does
    (num_cpus::get(), render.rs:129) — the affinity mask, cut to the cgroup's CPU quota when there is one (threads beyond
    the quota would only be throttled)."""
    n = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    return max(1, min(n, quota) if quota else n), n, quota


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


def cpu_baseline_run(flat, width, height, depth, seed, threads):
    """Time the CPU oracle on this host with `threads` worker threads: one full-image sample slice per thread
    (render_multi's own decomposition, render.rs:128-151), the process confined to as many cores as it runs threads,
    median of three runs of a few seconds each on a reduced image of the same scene and camera (throughput per sample
    does not depend on the image size: the pixels of the small image are a regular subsample of the large one's)."""
    import oracle_lib as O
    allowed = sorted(os.sched_getaffinity(0))
    before = set(allowed)
    try:
        os.sched_setaffinity(0, set(allowed[:threads]))  # the oracle's worker threads inherit the mask
    except OSError:
        pass
    w = min(width, 200)
    h = max(1, min(height, round(height * w / width)))

    def timed(spt):
        # (the reference's walk: every path to its end, dead or not, render.rs:29-34 -- it is the reference that is being timed)
        p = O.make_params(w, h, threads * spt, n_slices=threads, max_depth=depth, seed=seed, flags=128)
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
    return {
        "value": round(med[0], 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{w}x{h} (the {width}x{height} camera), {threads * spt} spp in {threads} slices, depth {depth}: "
                  f"{w * h * threads * spt} samples per run, median of 3 runs ({', '.join(f'{x[2]:.1f} s' for x in runs)})",
        "runs_msamples_per_s": [round(x[0], 4) for x in runs],
        "mrays_per_s": round(med[1], 3),
        "confined_to_cpus": allowed[:threads],
    }


def cpu_baseline(flat, width, height, depth, seed):
    """The CPU oracle on every core this process may use (the reference runs num_cpus::get() threads, render.rs:129);
    when that is more than 16, the 16-core figure of the earlier rounds is measured as well (`cpu_baseline_16`)."""
    threads, available, quota = host_threads()
    out = cpu_baseline_run(flat, width, height, depth, seed, threads)
    model, mhz = cpu_info()
    out.update({
        "cores_available": available, "cpu_quota": quota, "cpu_model": model, "cpu_mhz_median": mhz,
        "note": "C++ restatement of the reference, walking every path to its end as it does (RR_FLAG_TRACE_DEAD_PATHS), without its "
                "per-hit Material clones, Box<PDF> allocations or atlas mutex, "
                "so faster than the Rust binary would be: GPU/CPU ratios against it are conservative; one thread per core the "
                "process may use (affinity mask cut to the cgroup CPU quota), median of three runs",
    })
    extra = cpu_baseline_run(flat, width, height, depth, seed, 16) if threads > 16 else None
    return out, extra


def latest_profile(pattern):
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            return json.load(open(path)), os.path.basename(path)
        except Exception:
            continue
    return None, None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="final", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (of the job under strong scaling, per GPU under weak)")
    ap.add_argument("--slices", type=int, default=8, help="sample slices K, the reference's n_threads (of the job / per GPU)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default strong: ONE job whatever N (and, with N > 1, the weak job measured as a second figure)")
    ap.add_argument("--partition", default=None, choices=["slices", "rows", "both"],
                    help="default: slices when N divides K (every GPU the same samples of the whole image: balanced), rows otherwise")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--flags", type=int, default=0, help="RR_FLAG_* bits for the timed renders (kernel organisation)")
    ap.add_argument("--semantics", default="reference", choices=["reference", "intended", "book"],
                    help="intended / book: the labelled alternative semantics (RR_SEMANTICS_INTENDED, RR_SEMANTICS_BOOK), never the default")
    ap.add_argument("--bvh", default="reference", choices=["reference", "sah"],
                    help="sah: every BVH of the scene is a surface-area-heuristic tree, NOT the reference's (a labelled option for "
                         "--semantics intended: what the reference's tree costs next to a conventional one)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented counting passes")
    ap.add_argument("--no-weak", action="store_true", help="with N > 1: skip the second (weak-scaling) measurement")
    ap.add_argument("--inprocess", action="store_true", help="one process, N devices through rr_multi_* instead of one rank per GPU")
    ap.add_argument("--devices", default=None, help="--inprocess: comma-separated device list (a device may repeat), default 0..N-1")
    ap.add_argument("--reduce", default="auto", choices=["auto", "rccl", "peer"], help="--inprocess: how slice groups are summed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the ranks (gloo: --launch-check, and rehearsals with --share-gpus: images are staged through the host)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="REHEARSAL, not a measurement of N GPUs: rank r renders on GPU r %% (GPUs visible), so that the N-rank code runs on a "
                         "box with fewer GPUs; needs --backend gloo (RCCL refuses two ranks on one GPU); the line says n_gpus = GPUs used")
    ap.add_argument("--trace-dead-paths", action="store_true",
                    help="RR_FLAG_TRACE_DEAD_PATHS: walk paths on after their throughput has become exactly zero, as the reference does "
                         "(the default ends them there: same samples unless the rest of the walk meets a NaN)")
    ap.add_argument("--no-full-walk", action="store_true", help="skip the second figure (one step with RR_FLAG_TRACE_DEAD_PATHS)")
    ap.add_argument("--specialize", default="auto", choices=["auto", "always", "off"],
                    help="rr_scene_specialize after the upload (outside the timed region, like the upload): auto = only programs without BVH subtrees")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, count them with one all-reduce, print {n_gpus, n_ranks_seen} and exit: no rendering")
    return ap.parse_args(argv)


def visible_gpus(args):
    """Devices this process could use, counted without initialising the GPU runtime.  (--launch-check on a host without
    GPUs takes the count from RR_BENCH_ASSUME_DEVICES: it tests the launcher, nothing is rendered.)"""
    if args.launch_check and os.environ.get("RR_BENCH_ASSUME_DEVICES", "").isdigit():
        return int(os.environ["RR_BENCH_ASSUME_DEVICES"])
    import torch
    return torch.cuda.device_count()


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a torch.distributed.run child.  This
    parent makes no GPU call before or after (a process that has initialised the GPU must not be replaced or forked)."""
    import socket
    import subprocess
    have = visible_gpus(args)
    if args.share_gpus:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --share-gpus needs --backend gloo (RCCL cannot hold one GPU twice)")
        if have < 1:
            raise SystemExit("bench.py: --share-gpus still needs one GPU")
    elif have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible: refusing to measure fewer GPUs than asked")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("# bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(args, world, rank):
    import torch
    import torch.distributed as dist
    dist.init_process_group(args.backend)
    t = torch.ones(1)
    if args.backend == "nccl":
        t = t.cuda(int(os.environ.get("LOCAL_RANK", "0")))
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": args.gpus, "n_ranks_seen": int(t.item()), "world_size": dist.get_world_size()}))
    dist.barrier()
    dist.destroy_process_group()


def default_partition(world, n_slices):
    """Slices when the GPUs divide the K slices evenly: every GPU renders the same number of samples of the WHOLE image, so
    the shares cost the same (the reference's own decomposition, render.rs:130-140).  Bands of rows otherwise (bit for bit
    the one-GPU image, but rows differ in cost)."""
    return "slices" if n_slices % world == 0 else "rows"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.trace_dead_paths:
        args.flags |= 128  # RR_FLAG_TRACE_DEAD_PATHS
    in_world = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.inprocess:
        if in_world and world > 1:
            raise SystemExit("bench.py: --inprocess is one process; do not start it under torch.distributed.run")
        return run_inprocess(args)
    if args.gpus > 1 and not in_world:
        launch_ranks(args, argv)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has WORLD_SIZE={world}: refusing to report one as the other")
    if args.launch_check:
        return launch_check(args, world, rank)
    return run_ranks(args, world, rank)


def load_scene(args):
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import abi, host
    scene_name, W, H, spp, depth = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    builder, flat, _ = host.build_config_scene(scene_name, seed=args.seed, bvh=args.bvh)
    if args.semantics == "intended":
        flat.semantics = abi.SEMANTICS_INTENDED
    elif args.semantics == "book":
        flat.semantics = abi.SEMANTICS_BOOK
    return builder, flat, scene_name, W, H, spp, depth


def run_inprocess(args):
    """One process, N devices: the library's rr_multi_* entry points (what a Rust or C host calls)."""
    import numpy as np
    import ctypes as C
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import abi, device, multi
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if args.devices and args.gpus == 1:
        args.gpus = len(devices)
    if len(devices) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but --devices lists {len(devices)}")
    have = device.device_count()
    if have < len(set(devices)) or max(devices) >= have:
        raise SystemExit(f"bench.py: devices {devices} asked for, {have} GPU(s) visible: refusing to measure fewer GPUs than asked")
    builder, flat, scene_name, W, H, spp, depth = load_scene(args)
    K = args.slices
    scaling = args.scaling or "strong"
    n = len(devices)
    partition = args.partition or default_partition(n, K)
    job = multi.job_params(W, H, spp, K, depth, args.seed, n, scaling, flags=args.flags)
    split = {"slices": abi.SPLIT_SLICES, "rows": abi.SPLIT_ROWS, "both": abi.SPLIT_BOTH}[partition]
    mc = device.MultiContext(devices, reduce=args.reduce)
    mc.upload(flat)
    if args.specialize != "off":
        mc.specialize(args.specialize)
    for _ in range(args.warmup):
        r = mc.render(job, split)
    t0 = time.perf_counter()
    samples = rays = 0
    for _ in range(args.steps):
        r = mc.render(job, split)
        samples += r["counters"]["samples"]
        rays += r["counters"]["rays"]
    elapsed = time.perf_counter() - t0
    info = mc.info()
    bands, groups = multi.grid(n, partition)
    out = {
        "metric": "Msamples/s", "value": round(samples / elapsed / 1e6, 3), "unit": "Msamples/s", "n_gpus": len(set(devices)),
        "n_ranks_seen": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {W}x{H}, {job.samples_per_pixel} spp in {job.n_slices} slices, depth {depth}",
                   "scene": scene_name, "width": W, "height": H, "spp": job.samples_per_pixel, "slices": job.n_slices, "max_depth": depth,
                   "parallelism": f"in-process rr_multi over devices {devices}: {partition} ({bands} bands x {groups} slice groups), "
                                  f"slice groups summed by {info['reduce']}",
                   "seed": args.seed, "flags": args.flags, "semantics": args.semantics},
        "mrays_per_s": round(rays / elapsed / 1e6, 3),
        "note": "host buffers: every step includes the copy of the image (u8, resolved and linear f32) to the host",
        "roofline": {"bound": "valu", "achieved": None, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                     "note": "the per-kernel figures are those of the one-rank-per-GPU run (same kernels); this mode times the whole call"},
    }
    print(json.dumps(out))
    mc.close()


def run_ranks(args, world, rank):
    import numpy as np
    import torch
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import abi, build, device, multi

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    rehearsal = bool(args.share_gpus and world > 1)
    if rehearsal:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --share-gpus needs --backend gloo (RCCL cannot hold one GPU twice)")
        local_rank = local_rank % torch.cuda.device_count()
    elif args.backend != "nccl" and world > 1:
        raise SystemExit("bench.py: N GPUs are measured over RCCL (--backend nccl); gloo is for --launch-check and --share-gpus rehearsals")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dist = None
    n_ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        n_ranks_seen = dist.get_world_size()
    gpus_used = min(world, torch.cuda.device_count()) if rehearsal else world

    builder, flat, scene_name, W, H, spp, depth = load_scene(args)
    K = args.slices
    scaling = args.scaling or "strong"
    partition = args.partition or default_partition(world, K)
    ctx = device.Context(local_rank)
    info = ctx.upload(flat)
    if args.specialize != "off":
        try:
            info = ctx.specialize(args.specialize)
        except Exception as e:  # (no runtime compiler on this host: the interpreter renders, and the line says specialized: false)
            print(f"# rr_scene_specialize failed, rendering with the interpreter: {e}", file=sys.stderr, flush=True)
            info = ctx.info()
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the small tensors of the bookkeeping collectives live
    stream = torch.cuda.current_stream().cuda_stream
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, partition, extra_flags=0, steps=None, warmup=None):
        """W untimed and K timed steps of one job shared by the ranks -> dict of what rank 0 reports."""
        n_steps = args.steps if steps is None else steps
        n_warm = args.warmup if warmup is None else warmup
        job = multi.job_params(W, H, spp, K, depth, args.seed, world, scaling, flags=args.flags | extra_flags)
        p = multi.rank_share(job, rank, world, partition)
        comb = multi.Combiner(dist, world, partition, H, stage_through_host=rehearsal)
        n_sl, rows = p.slice_end - p.slice_begin, p.rows
        slice_sums = torch.empty((max(1, n_sl), rows, W, 3), dtype=torch.float32, device=dev)
        resolved = torch.empty((rows, W, 3), dtype=torch.float32, device=dev)
        linear = torch.empty((rows, W, 3), dtype=torch.float32, device=dev)
        counters = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)

        def step(ev=None):
            if ev:
                ev[0].record()
            ctx.render_device(p, slice_sums.data_ptr(), resolved.data_ptr(), linear.data_ptr(), counters.data_ptr(), stream)
            if ev:
                ev[1].record()
            whole = comb.combine((resolved, linear))
            if rank == 0:
                ctx.cast_u8_device(whole[0].data_ptr(), rgb8.data_ptr(), whole[0].numel(), stream)

        for _ in range(n_warm):
            step()
        barrier()
        counters.zero_()
        torch.cuda.synchronize()
        # HIP events on the stream the kernels are launched on (torch's current stream is the one handed to the library;
        # what the library runs on its second stream is ordered inside the bracket by its own events)
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(events[i])
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = [a.elapsed_time(b) for a, b in events]
        per_rank_ms = [float(np.mean(kernel_ms))]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            tot = counters.to(cdev, copy=True)
            dist.all_reduce(tot)
            counts_all = tot.cpu().numpy()
            km = torch.tensor(per_rank_ms, dtype=torch.float64, device=cdev)
            allk = [torch.empty_like(km) for _ in range(world)]
            dist.all_gather(allk, km)
            per_rank_ms = [float(x.item()) for x in allk]
        else:
            counts_all = counters.cpu().numpy()
        image = None
        if rank == 0:  # the job's 8-bit image as rank 0 holds it after the last step: what a check of an N-GPU run compares
            import hashlib
            img = rgb8.cpu().numpy()
            image = {"sha256": hashlib.sha256(img.tobytes()).hexdigest(), "mean_rgb": [round(float(x), 4) for x in img.reshape(-1, 3).mean(axis=0)]}
        spt = job.samples_per_pixel // job.n_slices
        samples_per_step = W * H * spt * job.n_slices
        assert int(counts_all[0]) == samples_per_step * n_steps, (int(counts_all[0]), samples_per_step, n_steps)
        return dict(job=job, comb=comb, elapsed=elapsed, kernel_ms=kernel_ms, per_rank_ms=per_rank_ms, counts_all=counts_all,
                    counts_rank=counters.cpu().numpy(), value=int(counts_all[0]) / elapsed / 1e6, image=image, steps=n_steps)

    m = measure(scaling, partition)
    # a second figure beside the headline: the same job with every dead path walked to its end, which is the work the reference
    # does (RR_FLAG_TRACE_DEAD_PATHS); one step, the kernels are warm
    full_walk = None
    if world == 1 and not (args.flags & abi.FLAG_TRACE_DEAD_PATHS) and not args.no_full_walk:
        full_walk = measure(scaling, partition, extra_flags=abi.FLAG_TRACE_DEAD_PATHS, steps=1, warmup=0)
    weak = None
    if world > 1 and args.scaling is None and not args.no_weak:
        weak = measure("weak", "slices")
    job, comb, elapsed, kernel_ms, counts_all, counts_rank = m["job"], m["comb"], m["elapsed"], m["kernel_ms"], m["counts_all"], m["counts_rank"]

    if rank == 0:
        total_samples = int(counts_all[0])
        total_rays = int(counts_all[1])
        value = m["value"]
        rays_per_launch = int(counts_rank[1]) / args.steps
        samples_per_launch = int(counts_rank[0]) / args.steps
        k_ms = float(np.mean(kernel_ms))
        pipeline = bool((args.flags & abi.FLAG_WAVEFRONT) or (info["wavefront"] and not (args.flags & abi.FLAG_PATHS)))
        if args.flags & abi.FLAG_PERSISTENT:
            kernel_name = "k_render (item-granular persistent kernel)"
        elif pipeline:
            kernel_name = "wavefront pipeline: k_wf_init + k_wf_traverse + k_wf_shade (camera rays) + k_wf_finish + k_wf_accumulate"
        else:
            kernel_name = ("k_wf_paths (persistent kernel at path granularity" + (", compiled for this scene's program by rr_scene_specialize" if info["specialized"] else "")
                           + ") + k_wf_accumulate")

        # What the kernels did per ray, from instrumented passes outside the timed region on a reduced image (per-ray
        # averages do not depend on the image size): `production` counts the tests of the traversal the timed kernels run
        # (redundant visits skipped, subtrees culled); `reference_order` counts every visit the reference's algorithm makes.
        per_ray = {}
        if not args.no_count:
            scale = min(1.0, (2.0e6 / (W * H * K * 4.0)) ** 0.5)
            Wc, Hc = max(16, int(W * scale)), max(16, int(H * scale))
            for label, flag in (("production", abi.FLAG_COUNT_PRODUCTION), ("reference_order", abi.FLAG_COUNTERS)):
                pc = device.make_params(Wc, Hc, K * 4, n_slices=K, max_depth=depth, seed=args.seed,
                                        flags=flag | (args.flags & (abi.FLAG_PERSISTENT | abi.FLAG_WAVEFRONT | abi.FLAG_PATHS | abi.FLAG_TRACE_DEAD_PATHS)))
                c2 = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)
                cs = torch.empty((K, Hc, Wc, 3), dtype=torch.float32, device=dev)
                ctx.render_device(pc, cs.data_ptr(), None, None, c2.data_ptr(), stream)
                torch.cuda.synchronize()
                per_ray[label] = dict(zip(COUNTER_NAMES, [int(x) for x in c2.cpu().numpy()]))
            print(f"# counting passes: {Wc}x{Hc}, {K * 4} spp", file=sys.stderr, flush=True)
        prod = per_ray.get("production")
        gather_bound = args.workload == "mesh"
        roofline = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                    "kernel": kernel_name, "kernel_ms": round(k_ms, 3), "rays_per_launch": int(rays_per_launch)}
        if prod:
            fpr, bpr = flops_per_ray(prod), bytes_per_ray(prod)
            achieved = fpr * rays_per_launch / (k_ms * 1e-3) / 1e12
            record_gbs = (bpr * rays_per_launch + 12.0 * samples_per_launch) / (k_ms * 1e-3) / 1e9
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
                "machine_lane_steps_per_ray": round(prod["machine_lane_steps"] / rays_c, 3),
                "record_bytes_per_ray": round(bpr, 1),
                "record_gbs": round(record_gbs, 1),
                "note": "bound = vector ALU issue: the scene records are cache-resident (scalar cache / L2; `record_gbs` is the "
                        "rate at which the kernels consume record bytes, served by caches, not HBM: see traffic and hbm_frac). "
                        "achieved = flops_per_ray (20/box test, 30/sphere, 15/rect, 60/triangle, +150 shading; tests counted on the "
                        "production traversal) x rays per launch / kernel time.  The parity contract forbids fused multiply-add "
                        "(-ffp-contract=off), so the attainable peak is peak_without_fma."})
            if gather_bound:
                # a scene larger than the L2s: every node and leaf record is a dependent random 32-64 byte read served by the
                # Infinity Cache (the scene fits its 256 MB) or HBM
                roofline.update({
                    "bound": "gather", "achieved": round(record_gbs, 1), "peak": GATHER_PEAK_GBS, "unit": "GB/s",
                    "frac": round(record_gbs / GATHER_PEAK_GBS, 5),
                    "valu": {"achieved_tflops": round(achieved, 3), "frac": round(achieved / VALU_PEAK_TFLOPS, 5)},
                    "note": "bound = random gather of scene records: achieved = record bytes per ray (32 B per box test, 16 B per sphere, "
                            "48 B per triangle; tests counted on the production traversal) x rays per launch + 12 B per sample, / kernel "
                            "time, against the Infinity-Cache random-row rate; hbm_frac is the measured HBM traffic against 8 TB/s"})
        # HBM bytes per launch and VALU issue figures from the rocprofv3 PMC passes of the same workload kept under profiles/
        digest = build.source_digest()
        tr, tr_src = latest_profile("r*_pmc_traffic.json")
        if tr and tr.get(args.workload, {}).get("hbm_bytes_per_sample") is not None:
            traffic = tr[args.workload]["hbm_bytes_per_sample"] * samples_per_launch
            roofline["traffic"] = round(traffic)
            roofline["hbm_frac"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roofline["traffic_source"] = tr_src
            roofline["traffic_stale"] = tr[args.workload].get("source_sha256") != digest
        summ, summ_src = latest_profile(f"r*_{args.workload}_summary.json")
        if summ:
            dd = summ.get("derived", {})
            roofline["measured"] = {k: dd.get(k) for k in ("kernel", "valu_issue_slot_frac", "active_lanes_per_valu_inst",
                                                          "valu_insts_per_wave_ray", "salu_insts_per_wave_ray", "waves_per_cu",
                                                          "lane_slot_frac") if k in dd}
            roofline["measured"]["source"] = summ_src
            roofline["measured"]["pmc_spp"] = summ.get("pmc_spp")
            # the counters were collected on the kernels whose sources hash to source_sha256: anything else is a stale figure
            roofline["measured"]["stale"] = summ.get("source_sha256") != digest
        out = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": gpus_used, "n_ranks_seen": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {job.samples_per_pixel} spp in {job.n_slices} slices, depth {depth}"
                                   + (f" ({spp} spp in {K} slices per GPU)" if scaling == "weak" and world > 1 else ""),
                       "scene": scene_name, "width": W, "height": H, "spp": job.samples_per_pixel, "slices": job.n_slices,
                       "max_depth": depth, "parallelism": f"{partition} x{world} ({comb.bands} bands x {comb.groups} slice groups)"
                                                           + (f"; REHEARSAL: {world} ranks share {gpus_used} GPU(s), images combined over gloo through the host"
                                                              if rehearsal else ""),
                       "seed": args.seed, "flags": args.flags, "semantics": args.semantics, "bvh": args.bvh,
                       "dead_paths": ("walked to their end, as the reference does (RR_FLAG_TRACE_DEAD_PATHS)" if args.flags & 128
                                      else "ended where their throughput becomes exactly zero (the library's default)"),
                       "stack_words": info["stack_words"],
                       "scene_bytes": info["device_bytes"], "specialized": info["specialized"],
                       "specialize_ms": round(info["specialize_ms"], 1)},
            "mrays_per_s": round(total_rays / elapsed / 1e6, 3),
            "rays_per_sample": round(total_rays / max(1, total_samples), 4),
            "kernel_ms_per_step": [round(x, 3) for x in kernel_ms],
            "kernel_ms_per_rank": [round(x, 3) for x in m["per_rank_ms"]],
            "source_sha256": digest,
            "image": m["image"],
            "roofline": roofline,
            "reference_published": {"msamples_per_s": 0.67, "source": "src/notes.txt:1-5 (Cornell box 600x600, 200 spp, depth 200, unknown CPU)"},
        }
        if full_walk:
            fc = full_walk["counts_all"]
            out["full_walk"] = {"value": round(full_walk["value"], 3), "unit": "Msamples/s", "steps": 1,
                                "ms_per_step": round(full_walk["elapsed"] * 1e3, 3),
                                "mrays_per_s": round(int(fc[1]) / full_walk["elapsed"] / 1e6, 3),
                                "rays_per_sample": round(int(fc[1]) / max(1, int(fc[0])), 4),
                                "note": "the same job with RR_FLAG_TRACE_DEAD_PATHS: every path is walked to its end after its throughput has become "
                                        "exactly zero, as the reference does (render.rs:29-34); `value` above is the library's default, which ends such "
                                        "paths (same samples unless the rest of the walk meets a NaN: tests/test_dead_paths.py)"}
        if weak:
            wj = weak["job"]
            out["weak"] = {"value": round(weak["value"], 3), "unit": "Msamples/s", "ms_per_step": round(weak["elapsed"] / args.steps * 1e3, 3),
                           "workload": f"{args.workload}: {W}x{H}, {wj.samples_per_pixel} spp in {wj.n_slices} slices ({spp} spp in {K} slices per GPU)",
                           "partition": "slices", "kernel_ms_per_rank": [round(x, 3) for x in weak["per_rank_ms"]]}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"], extra = cpu_baseline(flat, W, H, depth, args.seed)
            if extra:
                out["cpu_baseline_16"] = extra
            out["speedup_vs_cpu_baseline"] = round(value / max(1e-9, out["cpu_baseline"]["value"]), 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
