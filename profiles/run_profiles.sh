#!/bin/bash
# Collects the rocprofv3 evidence for one workload on the GPU box:
#   1. kernel trace + stats of a bench.py run (the kernels' durations must agree with bench.py's own
#      HIP-event timing);
#   2. PMC passes, one rocprofv3 run each (counters are not split over passes automatically), on a
#      one-step run of the same workload.
# Usage (from the repo root, on the GPU box):  bash profiles/run_profiles.sh <tag> [workload] [spp] [spp of the PMC passes]
# (spp 0 = the workload's own: the trace pass is then bench.py's default command for that workload)
# Output goes to gpurun_out/prof_<tag>_<workload>/; profiles/summarize.py turns it into profiles/<tag>_<workload>_*.
set -u
TAG=${1:-r2}
WORKLOAD=${2:-cornell}
SPP=${3:-0}
PMC_SPP=${4:-$SPP}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_$WORKLOAD
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$ROOT/bench.py" --workload "$WORKLOAD" --spp "$SPP" --steps 3 --warmup 1 --no-cpu --no-count --no-full-walk > "$OUT/bench_under_trace.json" 2> "$OUT/bench_under_trace.err" || { echo "trace pass failed"; exit 1; }
echo "trace ok"

pass() {
    name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
        python3 "$ROOT/bench.py" --workload "$WORKLOAD" --spp "$PMC_SPP" --steps 1 --warmup 0 --no-cpu --no-count --no-full-walk > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || { echo "pmc pass $name failed"; exit 1; }
    echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH &&
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 &&
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 &&
pass tcc1 FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE GRBM_COUNT &&
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass occ MeanOccupancyPerCU &&
pass valu_util VALUUtilization &&
pass valu_busy VALUBusy
echo "profiles done: $OUT"
