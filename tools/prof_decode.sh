#!/bin/bash
# rocprofv3 of the beam-search kernel alone, this round's library vs the round-1 decoder (tools/variants/libradian_hip_r1decode.so:
# round 1's decode.hip linked with the current library), same inputs: kernel trace + one SQ PMC pass per configuration.
# usage: tools/prof_decode.sh <tag> ["new old"]      (on the GPU box via gpurun)
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profdec_$TAG
mkdir -p $OUT
cd $R
ARMS=${2:-"new old"}       # second argument: which arms to run ("new", "old" or both)
for LIBTAG in $ARMS; do
  if [ $LIBTAG = old ]; then export RADIAN_HIP_LIB=${OLD_LIB:-$R/tools/variants/libradian_hip_r1decode.so}; else unset RADIAN_HIP_LIB; fi
  CFGS=("512 10 0" "512 10 1" "4096 10 0" "512 25 1" "4096 25 1")
  # this round's decoder also in its default arithmetic (glibc's operation sequence; the five above run the fast routines)
  if [ $LIBTAG = new ] || [ -n "${OLD_HAS_GLIBC:-}" ]; then CFGS+=("512 10 0 glibc" "512 10 1 glibc" "4096 10 0 glibc" "512 25 1 glibc"); fi
  # round 3: the reference's default width as one and as two sequences per wave (this round's library only)
  if [ $LIBTAG = new ] && [ -n "${W6:-}" ]; then CFGS=("4096 6 0 glibc one" "4096 6 0 glibc two" "4096 6 1 glibc one" "4096 6 1 glibc two" "512 6 1 glibc one" "512 6 1 glibc two"); fi
  # round 4: the metric's width as one and as two sequences per wave (two candidates per lane of a half-wave)
  if [ $LIBTAG = new ] && [ -n "${W10:-}" ]; then CFGS=("4096 10 0 glibc one" "4096 10 0 glibc two" "4096 10 1 glibc one" "4096 10 1 glibc two" "4096 10 0 fast one" "4096 10 0 fast two"); fi
  for CFG in "${CFGS[@]}"; do
    NAME=${LIBTAG}_$(echo $CFG | tr ' ' '_')
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 tools/decode_prof_run.py $CFG > $OUT/trace_$NAME.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_$NAME -- python3 tools/decode_prof_run.py $CFG > $OUT/pmc_$NAME.log 2>&1
    echo "done $NAME"
  done
done
