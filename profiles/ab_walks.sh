#!/bin/bash
# Both walks of dead paths on every path-kernel workload: bash profiles/ab_walks.sh "<workloads>" <spp>
WORKLOADS=$1; SPP=$2
for w in $WORKLOADS; do
  for extra in "" "--trace-dead-paths"; do
    out=$(timeout -k 10 250 python bench.py --workload $w --spp $SPP --steps 3 --warmup 1 --no-cpu --no-full-walk $extra 2>gpurun_out/ab_err.log) || { echo "FAILED: $w $extra"; tail -5 gpurun_out/ab_err.log; exit 1; }
    echo "$w ${extra:-default} $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], "Mrays/s", d["mrays_per_s"], "rays/sample", d["rays_per_sample"], "frac", r["frac"], "flops/ray", r.get("flops_per_ray"), "spec", d["config"]["specialized"])')"
  done
done
