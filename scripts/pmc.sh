# usage: bash scripts/pmc.sh <outdir-under-gpurun_out> [bench args...]
# PMC passes on the bench command (own runs, --kernel-trace only), per MI355X_MICROARCH.md; the kernel-trace pass
# profiles 60 launches (20 of them warm-up) so that its average is a warm one
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --settle-ms 0 --cpu-sample 0 --no-latency $@"   # (counters do not depend on the clocks: no settle launches)
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 20 --cpu-sample 0 --no-latency $@ > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -o p -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -o p -- $CMD > $OUT/p5.log 2>&1
find $OUT -name "*.csv" | head -30
