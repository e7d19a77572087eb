# one-rank RCCL group (the N > 1 code path on one GPU) against the plain single-GPU step, same box
for i in 1 2; do
for m in plain force force_z1; do
  unset XGGM_DP_FORCE; Z=""
  if [ $m != plain ]; then export XGGM_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577; fi
  if [ $m = force_z1 ]; then Z="--zero1 1"; fi
  python bench.py $Z --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$m', d['ms_per_step'], d['ms_per_pass'])
"
done
done
