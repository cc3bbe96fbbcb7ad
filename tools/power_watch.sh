#!/bin/bash
# Shader clock and socket power while the default bench loop runs (is the conv kernel power-bound?): samples rocm-smi twice a
# second beside `bench.py --steps N`.  usage: tools/power_watch.sh [steps] [extra bench args]   (on the GPU box)
STEPS=${1:-1500}
shift
OUT=gpurun_out/power_watch.txt
: > $OUT
( for i in $(seq 1 60); do
    echo "== t=$i" >> $OUT
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $OUT
    sleep 0.5
  done ) &
W=$!
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps $STEPS --warmup 20 "$@" > gpurun_out/power_watch_bench.json 2> gpurun_out/power_watch_bench.err
RC=$?
kill $W 2>/dev/null
wait $W 2>/dev/null
python3 - <<'PY'
import json, re
d = json.loads(open("gpurun_out/power_watch_bench.json").read().strip().splitlines()[-1])
print("bench: %.2f M samples/s, %.3f ms/step, conv %.4f ms/launch" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
t = open("gpurun_out/power_watch.txt").read()
sclk = [int(x) for x in re.findall(r"sclk.*?\((\d+)Mhz\)", t)]
pw = [float(x) for x in re.findall(r"Power.*?:\s*([0-9.]+)", t)]
print("sclk samples (MHz):", sclk)
print("power samples (W):", pw)
PY
exit $RC
