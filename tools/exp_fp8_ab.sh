# same-box A/B: bf16 step vs fp8-forward step (BASELINE configs[4]); XGGM_FP8_8W=0: without the 8-wave e4m3 tile
for i in 1 2 3; do
for d in bf16 fp8 fp8no8; do
  if [ $d = fp8no8 ]; then export XGGM_FP8_8W=0; dd=fp8; else unset XGGM_FP8_8W; dd=$d; fi
  python bench.py --dtype $dd --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$d', d['ms_per_step'], d['ms_per_pass'])
"
done
done
