for a in "--steps 20" "--steps 200" "--steps 200 --decode-group 4" "--steps 200 --decode-group 16" "--steps 200 --decode-math fast" "--steps 20 --decode-group 4" "--steps 20 --decode-group 2" "--steps 200 --lanes 3"; do
  echo "== $a"; python bench.py --no-secondary --no-cpu-baseline $a 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('frac'), j['roofline'].get('pipeline_frac'))"
done
