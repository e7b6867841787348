# usage: trace_quick.sh <workload> <spp>   — kernel trace with per-kernel statistics of one bench step
set -u
W=${1:-suzanne}; SPP=${2:-200}
ROOT=/root/repo; OUT=$ROOT/gpurun_out/trace_$W; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 0 --spp $SPP --no-cpu > $OUT/bench.json 2> $OUT/bench.err || { echo trace failed; tail -5 $OUT/bench.err; exit 1; }
find $OUT -name '*kernel_stats.csv' | head -1 | xargs cat | cut -c1-200
