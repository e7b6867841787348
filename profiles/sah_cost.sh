#!/bin/bash
# What the reference's BVH (median split of a random axis, collider.rs:173-212) costs next to a surface-area-heuristic tree, under
# the labelled intended semantics, where a BVH only accelerates: bash profiles/sah_cost.sh  (on the GPU box)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out/sah"
for w in "final 1000" "suzanne 1000" "spheres 1000" "mesh 8"; do
    set -- $w
    for bvh in reference sah; do
        timeout -k 10 300 python3 "$ROOT/bench.py" --workload "$1" --spp "$2" --steps 3 --warmup 1 --no-cpu --semantics intended --bvh $bvh \
            > "$ROOT/gpurun_out/sah/$1_$bvh.json" 2> "$ROOT/gpurun_out/sah/$1_$bvh.err" || { echo "$1 $bvh failed"; tail -3 "$ROOT/gpurun_out/sah/$1_$bvh.err"; exit 1; }
        python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[2], sys.argv[3], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'rays/sample', d['rays_per_sample'], r.get('per_ray_production'))" "$ROOT/gpurun_out/sah/$1_$bvh.json" "$1" "$bvh"
    done
done
