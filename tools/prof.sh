#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box via gpurun): kernel trace + PMC passes (separate runs).
# usage: tools/prof.sh <tag>
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
ARGS="bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary --lanes 1"
# kernel trace twice: the default command (two forward lanes: launches of consecutive batches overlap, so a launch's
# duration includes the time it shares the chip with the other lane's launch) and one lane (launches back to back: the
# per-launch durations bench.py's roofline leg measures with HIP events on its single-stream timing pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/trace2_bench.json 2> $OUT/trace2_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace_bench.json 2> $OUT/trace_err.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq_err.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write_bench.json 2> $OUT/pmc_write_err.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- python3 $ARGS > $OUT/pmc_lds_bench.json 2> $OUT/pmc_lds_err.log
find $OUT -name "*.csv" | head -30
