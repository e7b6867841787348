run() { out=$(RR_HIP_LIB=$1 timeout -k 10 200 python bench.py "${@:2}" --steps 3 --warmup 1 --no-cpu --no-count --no-full-walk 2>gpurun_out/ab_err.log) || { echo FAILED; tail -3 gpurun_out/ab_err.log; exit 1; }
 echo "${*:2} ${1:-prod} $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["image"]["sha256"][:12])')"; }
run "" --workload mesh --spp 8; for w in suzanne final spheres; do run "" --workload $w --spp 64 --flags 4; done
