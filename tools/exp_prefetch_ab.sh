# same-box A/B of the weight prefetch beside the LayerNorm launches (XGGM_PREFETCH=0 turns it off), alternating
for i in 1 2 3; do
for v in off on; do
  if [ $v = off ]; then export XGGM_PREFETCH=0; else unset XGGM_PREFETCH; fi
  python bench.py --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('prefetch $v', d['ms_per_step'], d['ms_per_pass'])
"
done
done
