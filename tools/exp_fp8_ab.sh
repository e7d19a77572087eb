# same-box A/B: bf16 step vs fp8-forward step (BASELINE configs[4]), alternating; optional args go to bench.py (--batch 92)
for i in 1 2 3; do
for d in bf16 fp8; do
  python bench.py --dtype $d --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$d', d['ms_per_step'], d['ms_per_pass'])
"
done
done
