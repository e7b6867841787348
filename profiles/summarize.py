#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>_<workload>/ (written by run_profiles.sh) into the files kept under profiles/:

  <tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, as written by the tool
  <tag>_<workload>_summary.json       bench.py's line under the trace, per-kernel time of the timed steps, PMC totals of the
                                      timed step's kernels and figures derived from them
  <tag>_pmc_traffic.json              HBM bytes per sample and workload (read by bench.py for roofline.traffic)

Only the kernels of the timed step count: bench.py's instrumented pass afterwards is cut off at the first dispatch of a
counting kernel (first template argument 1 or 2).

usage: python profiles/summarize.py <tag> <workload>
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{workload}")
dst = os.path.join(ROOT, "profiles")


KERNEL = re.compile(r"\b(k_render|k_wf_paths|k_wf_traverse|k_wf_finish)<(\d)")  # first template argument: 0 = production, 1 / 2 = counting builds


def short(name):
    m = KERNEL.search(name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    for k in ("k_wf_shade", "k_wf_init", "k_wf_accumulate", "k_resolve", "k_cast_u8"):
        if k in name:
            return k
    return None


def newest(pattern):
    """gpurun merges every run into the same local directory: take the latest file only."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


def counting(name):
    m = KERNEL.search(name)
    return bool(m) and m.group(2) != "0"


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


out = {"tag": tag, "workload": workload}
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv"))
bench = last_json_line(os.path.join(src, "bench_under_trace.json"))
out["bench_under_trace"] = bench
out["source_sha256"] = bench.get("source_sha256")  # the device sources the profiled kernels were built from (bench.py compares)

# per-kernel time of the timed steps (the warm-up step and the instrumented pass excluded)
trace = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if trace:
    rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if short(r["Kernel_Name"])]
    cut = next((i for i, r in enumerate(rows) if counting(r["Kernel_Name"])), len(rows))
    rows = rows[:cut]
    steps = bench["steps"] + bench["warmup"]
    casts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_cast_u8"]
    if casts:
        rows = rows[:casts[-1] + 1]  # a timed step ends with its cast: what follows belongs to the instrumented pass
    if len(casts) == steps and bench["warmup"] > 0:
        rows = rows[casts[bench["warmup"] - 1] + 1:]  # drop the warm-up steps
    per = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        e = per.setdefault(k, {"launches": 0, "total_ms": 0.0})
        e["launches"] += 1
        e["total_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for e in per.values():
        e["launches_per_step"] = e["launches"] / bench["steps"]
        e["ms_per_step"] = round(e["total_ms"] / bench["steps"], 4)
        e["total_ms"] = round(e["total_ms"], 4)
    out["kernel_time_timed_steps"] = per
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6 / bench["steps"] if rows else None
    out["gpu_span_ms_per_step"] = round(span, 3) if span else None
    out["bench_hip_event_ms_per_step"] = bench["roofline"]["kernel_ms"]

# PMC: totals over the timed step's dispatches, per kernel
pmc = collections.OrderedDict()
pmc_bench = None
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = newest(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    if pmc_bench is None:
        pmc_bench = last_json_line(d + ".json")
    rows = [r for r in csv.DictReader(open(files[0])) if short(r["Kernel_Name"])]
    ids = [int(r["Dispatch_Id"]) for r in rows if counting(r["Kernel_Name"])]
    cut = min(ids) if ids else 1 << 62
    for r in rows:
        if int(r["Dispatch_Id"]) >= cut:
            continue
        pmc.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], 0.0)
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
out["pmc_timed_step"] = pmc
if pmc_bench:
    out["bench_under_pmc"] = {k: pmc_bench[k] for k in ("value", "ms_per_step", "mrays_per_s", "config")}
    out["pmc_spp"] = pmc_bench["config"]["spp"]
    if pmc_bench.get("source_sha256") != out["source_sha256"]:
        raise SystemExit("the trace pass and the PMC passes ran different device sources")
    rays = pmc_bench["roofline"]["rays_per_launch"]
    samples = pmc_bench["value"] * 1e6 * pmc_bench["ms_per_step"] * 1e-3
    tot = collections.Counter()
    for k, c in pmc.items():
        if k in ("k_resolve", "k_cast_u8"):
            continue
        for name, v in c.items():
            tot[name] += v
    derived = {"rays": rays, "samples": round(samples)}
    if "SQ_INSTS_VALU" in tot:
        derived["valu_insts_per_wave_ray"] = round(tot["SQ_INSTS_VALU"] / (rays / 64.0), 1)
        derived["salu_insts_per_wave_ray"] = round(tot["SQ_INSTS_SALU"] / (rays / 64.0), 1)
        derived["vmem_rd_insts_per_wave_ray"] = round(tot["SQ_INSTS_VMEM_RD"] / (rays / 64.0), 2)
        derived["lds_insts_per_wave_ray"] = round(tot["SQ_INSTS_LDS"] / (rays / 64.0), 2)
    k_ms = pmc_bench["roofline"]["kernel_ms"]  # HIP events around the path-tracing kernels of the step the counters cover
    main = max((k for k in pmc if k not in ("k_resolve", "k_cast_u8")), key=lambda k: pmc[k].get("SQ_INSTS_VALU", 0.0), default=None)
    derived["kernel"] = main
    if "SQ_INSTS_VALU" in tot and k_ms:
        # a SIMD issues one wave64 vector instruction every 2 cycles (two passes of 32 lanes): 1024 SIMDs x 1.2 G/s at 2.4 GHz
        derived["valu_issue_slot_frac"] = round(tot["SQ_INSTS_VALU"] / (1024 * 1.2e9 * k_ms * 1e-3), 4)
    if "SQ_THREAD_CYCLES_VALU" in tot and tot.get("SQ_ACTIVE_INST_VALU"):
        # both counters tick in units of 4 cycles; the ratio is the mean number of active lanes per VALU instruction
        derived["active_lanes_per_valu_inst"] = round(tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"], 2)
    if "valu_issue_slot_frac" in derived and "active_lanes_per_valu_inst" in derived:
        derived["lane_slot_frac"] = round(derived["valu_issue_slot_frac"] * derived["active_lanes_per_valu_inst"] / 64.0, 4)
    if tot.get("SQ_WAVE_CYCLES"):
        derived["wave_cycles_split"] = {k: round(tot[k] / tot["SQ_WAVE_CYCLES"], 4)
                                        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in tot}
    if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
        # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled
        fetch = tot.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        write = tot.get("WRITE_SIZE", 0.0) * 1024.0
        derived["hbm_fetch_bytes"] = round(fetch)
        derived["hbm_write_bytes"] = round(write)
        derived["hbm_bytes_per_sample"] = (fetch + write) / max(1.0, samples)
        tpath = os.path.join(dst, f"{tag}_pmc_traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t[workload] = {"hbm_bytes_per_sample": derived["hbm_bytes_per_sample"], "hbm_fetch_bytes": round(fetch),
                       "hbm_write_bytes": round(write), "samples": round(samples), "source_sha256": out["source_sha256"],
                       "pmc_spp": out.get("pmc_spp"),
                       "note": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte reads at 64 bytes) + WRITE_SIZE x 1024, "
                               "summed over the timed step's path-tracing kernels; Infinity-Cache hits are included"}
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
        if k_ms:
            derived["hbm_gbs"] = round((fetch + write) / (k_ms * 1e-3) / 1e9, 1)
    if tot.get("TCC_HIT_sum") is not None and tot.get("TCC_MISS_sum") is not None and (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]) > 0:
        derived["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 4)
    if tot.get("TCC_REQ_sum") and k_ms:
        derived["l2_requests_per_ray"] = round(tot["TCC_REQ_sum"] / rays, 2)
    if tot.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        derived["l1_accesses_per_ray"] = round(tot["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays, 2)
        if tot.get("TCP_TCC_READ_REQ_sum") is not None:
            derived["l1_read_miss_per_ray"] = round(tot["TCP_TCC_READ_REQ_sum"] / rays, 2)
    # per kernel: instructions, lanes and wait split, so that a pipeline of several kernels can be read kernel by kernel
    per_kernel = {}
    for k, c in pmc.items():
        e = {}
        if c.get("SQ_ACTIVE_INST_VALU"):
            e["active_lanes_per_valu_inst"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"], 2)
        if c.get("SQ_WAVE_CYCLES"):
            e["wave_cycles_split"] = {n: round(c[n] / c["SQ_WAVE_CYCLES"], 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA") if n in c}
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES"):
            if n in c:
                e[n] = c[n]
        if "FETCH_SIZE" in c:
            e["hbm_fetch_bytes"] = round(c["FETCH_SIZE"] * 2048.0)
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes"] = round(c["WRITE_SIZE"] * 1024.0)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if e:
            per_kernel[k] = e
    derived["per_kernel"] = per_kernel
    # percentages are per dispatch: report their mean over the dispatches of each kernel
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        files = newest(os.path.join(d, "*", "*_counter_collection.csv")) if os.path.isdir(d) else []
        for f in files:
            acc = collections.defaultdict(list)
            rows = [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"])]
            ids = [int(r["Dispatch_Id"]) for r in rows if counting(r["Kernel_Name"])]
            cut = min(ids) if ids else 1 << 62
            for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
                if r["Counter_Name"] in ("MeanOccupancyPerCU", "VALUUtilization", "VALUBusy") and int(r["Dispatch_Id"]) < cut:
                    acc[(r["Counter_Name"], short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
            for (name, k), v in acc.items():
                derived.setdefault(name + "_mean_per_dispatch", {})[k] = round(sum(v) / len(v), 3)
                # the first dispatch of a wavefront kernel is the largest (every path of the batch is alive): the mean over
                # dispatches is dominated by the many near-empty launches at the end of a bounce loop
                derived.setdefault(name + "_first_dispatch", {})[k] = round(v[0], 3)
    if main and "MeanOccupancyPerCU_mean_per_dispatch" in derived:
        derived["waves_per_cu"] = derived["MeanOccupancyPerCU_mean_per_dispatch"].get(main)
    out["derived"] = derived
json.dump(out, open(os.path.join(dst, f"{tag}_{workload}_summary.json"), "w"), indent=1)
print(json.dumps(out.get("derived", {}), indent=1))
print(json.dumps(out.get("kernel_time_timed_steps", {}), indent=1))
