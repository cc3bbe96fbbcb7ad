#!/bin/bash
# Beam search alone under rocprofv3: a --kernel-trace --stats pass and a SEPARATE --pmc pass per configuration (the two are never combined).
# usage (through gpurun): bash tools/prof_decode.sh <out dir under gpurun_out> "<n windows> <W> <soft 0|1> <math> [form]" ...
# then: python3 tools/prof_decode_summary.py gpurun_out/<dir> <tag>   ->  profiles/<tag>_beam_search_pmc.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    name=new_$(echo $cfg | tr ' ' '_')
    # (rocprofv3 may fault at exit on this image after it has written its csv files: not fatal)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- python3 "$ROOT/tools/decode_prof_run.py" $cfg > "$OUT/trace_$name.log" 2>&1 || true
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/tools/decode_prof_run.py" $cfg > "$OUT/pmc_$name.log" 2>&1 || true
    grep timesteps_per_launch "$OUT/trace_$name.log" || true
done
