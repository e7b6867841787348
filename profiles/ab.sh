#!/bin/bash
# A/B of builds of the device library on the GPU box: bash profiles/ab.sh "<workloads>" <spp> <lib> [<lib> ...]
# ("" or "prod" = the in-tree production build).  Each run is bounded; the script stops at the first failure.
WORKLOADS=$1; SPP=$2; shift 2
for w in $WORKLOADS; do
  for lib in "$@"; do
    [ "$lib" = prod ] && lib=""
    out=$(RR_HIP_LIB=$lib timeout -k 10 150 python bench.py --workload $w --spp $SPP --steps 3 --warmup 1 --no-cpu --no-count --no-full-walk 2>gpurun_out/ab_err.log) || { echo "FAILED: $w $lib"; tail -5 gpurun_out/ab_err.log; exit 1; }
    echo "$w ${lib:-prod} $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
