for a in "--steps 200 --decode-form one" "--steps 200 --decode-form two" "--steps 200 --decode-form one" "--steps 200 --decode-form two" "--steps 200 --decode-form one --decode-math fast" "--steps 200 --decode-form two --decode-math fast"; do
  echo "== $a"; python bench.py --no-secondary --no-cpu-baseline $a 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('frac'), j['roofline'].get('pipeline_frac'), j.get('decode_timesteps_per_s'))"
done
