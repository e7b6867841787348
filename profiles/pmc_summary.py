#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs for one kernel and normalises per wave-bounce (64 rays)."""
import collections, csv, glob, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmcq'
kern = sys.argv[2] if len(sys.argv) > 2 else 'k_render<false>'
res = collections.OrderedDict()
for f in sorted(glob.glob(d + '/*/vm/*_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if kern in row['Kernel_Name']:
            res.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
rays = ms = None
for j in sorted(glob.glob(d + '/*.json')):
    try:
        b = json.load(open(j)); rays = b['roofline']['rays_per_launch']; ms = b['roofline']['kernel_ms']; break
    except Exception:
        pass
print('rays per launch', rays, 'kernel ms', ms)
for k, v in res.items():
    print(f"{k:28s} n={len(v)} total {sum(v):.4g}" + (f"  per wave-bounce {sum(v)/len(v)/(rays/64):.1f}" if rays else ""))
