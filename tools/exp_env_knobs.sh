# runtime knobs that touch the launch path, same box, alternating: ms per iteration of the resident leg
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'])
"; }
for i in 1 2; do
  run default
  HIP_FORCE_DEV_KERNARG=1 run dev_kernarg_1
  HIP_FORCE_DEV_KERNARG=0 run dev_kernarg_0
  GPU_MAX_HW_QUEUES=1 run one_hw_queue
  HSA_NO_SCRATCH_RECLAIM=1 run no_scratch_reclaim
done
