for f in 0 1 0 1; do RD_FUSE=$f python tools/fwd_lanes.py reads 2 80 1 | tail -1; done
for f in 0 1 0 1; do RD_FUSE=$f python tools/fwd_lanes.py reads 1 80 1 | tail -1; done
for a in "--conv-fuse 0" "--conv-fuse 1" "--conv-fuse 0" "--conv-fuse 1"; do
  echo "== $a"; python bench.py --no-secondary --no-cpu-baseline --steps 200 $a 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('frac'), j['roofline'].get('pipeline_frac'))"
done
