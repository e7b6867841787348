#!/bin/bash
# The per-scene compile (rr_scene_specialize) against the interpreter, same library: bash profiles/ab_specialize.sh "<workloads>" <spp>
WORKLOADS=$1; SPP=$2
for w in $WORKLOADS; do
  for mode in off always; do
    out=$(timeout -k 10 200 python bench.py --workload $w --spp $SPP --steps 3 --warmup 1 --no-cpu --no-count --no-full-walk --specialize $mode 2>gpurun_out/ab_err.log) || { echo "FAILED: $w $mode"; tail -5 gpurun_out/ab_err.log; exit 1; }
    echo "$w $mode $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["specialized"], d["config"]["specialize_ms"])')"
  done
done
