# Latency-side counters of the strip kernel (separate passes, <= 8 SQ counters each): average LDS / vector-memory latency
# = SQ_INST_LEVEL_x / SQ_INSTS_x, LDS queue back-pressure, scalar / branch / instruction-fetch cycles.
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/tools/strip_bench.py --shape reddit --F 256 --max-ratio 0.7 --piece 0 --block 52 --iters 2 --no-base"
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_strip2.txt
: > $out
rm -rf /tmp/pmc_a && timeout -s KILL 150 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_a -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_strip2_run.log 2>&1 && python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_a spmm_strip >> $out || exit 1
rm -rf /tmp/pmc_b && timeout -s KILL 150 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_b -- $B >> $GRAFT_REPO_ROOT/gpurun_out/pmc_strip2_run.log 2>&1 && python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_b spmm_strip >> $out || exit 2
