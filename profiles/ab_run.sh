#!/bin/bash
# A/B on the GPU box: bash profiles/ab_run.sh <workload> <spp> <variant> [<variant> ...]
# ("base" = the in-tree library).  One bench line per variant (no CPU leg, no counting passes), value and ms per step to stdout.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$1; SPP=$2; shift 2
mkdir -p "$ROOT/gpurun_out/ab"
for v in "$@"; do
    if [ "$v" = base ]; then unset RR_HIP_LIB; else export RR_HIP_LIB=$ROOT/ab/librr_hip_$v.so; fi
    timeout -k 10 300 python3 "$ROOT/bench.py" --workload "$W" --spp "$SPP" --steps 3 --warmup 1 --no-cpu --no-count ${AB_FLAGS:-} \
        > "$ROOT/gpurun_out/ab/${W}_$v.json" 2> "$ROOT/gpurun_out/ab/${W}_$v.err" || { echo "$v failed"; tail -5 "$ROOT/gpurun_out/ab/${W}_$v.err"; exit 1; }
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['rays_per_sample'])" "$ROOT/gpurun_out/ab/${W}_$v.json" "$W/$v"
done
