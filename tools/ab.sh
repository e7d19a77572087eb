#!/bin/bash
# same-box A/B of the product library against an experimental build (make -C x-ggm_amd/csrc alt ALT="-D..."):
# alternates the two libraries N times and prints ms per iteration of each run.  Run ON the GPU box.
N=${1:-3}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for i in $(seq $N); do
  for v in base alt; do
    if [ $v = alt ]; then export XGGM_LIB="$ROOT/x-ggm_amd/csrc/build_alt/libxggm_hip.so"; else unset XGGM_LIB; fi
    python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['ms_per_pass'])
"
  done
done
