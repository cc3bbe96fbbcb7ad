#!/bin/bash
# quick profile of the bench's one-lane forward: kernel trace + LDS PMC pass.  usage: tools/prof_quick.sh <tag> [precision]
set -u
TAG=${1:-q}
PREC=${2:-fp32}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profq_$TAG
mkdir -p $OUT
cd $R
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --lanes 1 --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace_bench.json 2> $OUT/trace_err.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- python3 $ARGS > $OUT/pmc_lds_bench.json 2> $OUT/pmc_lds_err.log
python3 - <<PY
import csv, glob, collections
st = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(st)):
    if float(r["Percentage"]) > 0.5: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob("$OUT/pmc_lds/*/*_counter_collection.csv")[0])):
    d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
