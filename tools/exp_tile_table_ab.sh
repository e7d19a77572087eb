# same-box A/B: the measured tile table on / off / a candidate file (XGGM_TILE_TABLE=path)
R="$(cd "$(dirname "$0")/.." && pwd)"
for i in 1 2 3; do
for m in table none ${1:-}; do
  if [ $m = none ]; then export XGGM_TILE_TABLE=0; elif [ $m = table ]; then unset XGGM_TILE_TABLE; else export XGGM_TILE_TABLE="$R/$m"; fi
  python bench.py --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$m', d['ms_per_step'], d['ms_per_pass'])
"
done
done
