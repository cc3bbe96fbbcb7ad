export RADIAN_HIP_LIB=$PWD/tools/variants/libradian_x.so
for e in "X=1" "RD_X_NO_DECODE=1" "X=1" "RD_X_NO_DECODE=1"; do
  echo "== $e"; env $e python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'])"
done
