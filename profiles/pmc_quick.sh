set -u
ROOT=/root/repo; OUT=$ROOT/gpurun_out/pmcq; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --spp 480 --no-cpu > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; exit 1; }; echo pass $name ok; }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH &&
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 &&
pass vu VALUUtilization && pass vb VALUBusy
