#!/bin/bash
# Same library, different RR_FLAG_* bits: bash profiles/ab_flags.sh <workload> <spp> <flags> [<flags> ...]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$1; SPP=$2; shift 2
mkdir -p "$ROOT/gpurun_out/ab"
for f in "$@"; do
    timeout -k 10 300 python3 "$ROOT/bench.py" --workload "$W" --spp "$SPP" --steps 3 --warmup 1 --no-cpu --no-count --flags "$f" \
        > "$ROOT/gpurun_out/ab/${W}_f$f.json" 2> "$ROOT/gpurun_out/ab/${W}_f$f.err" || { echo "flags $f failed"; tail -5 "$ROOT/gpurun_out/ab/${W}_f$f.err"; exit 1; }
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], 'Msamples/s', d['ms_per_step'], 'ms')" "$ROOT/gpurun_out/ab/${W}_f$f.json" "$W/flags=$f"
done
