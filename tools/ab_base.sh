#!/bin/bash
# same-box A/B of the working tree against a base checkout in _base/ (git worktree add _base <commit>; make -C
# _base/x-ggm_amd/csrc): alternates the two N times and prints ms per iteration of each run.  Run ON the GPU box.
N=${1:-3}
shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for i in $(seq $N); do
  for v in base new; do
    if [ $v = base ]; then D="$ROOT/_base"; else D="$ROOT"; fi
    (cd "$D" && python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null) | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d.get('ms_per_pass'), (d.get('value_with_loader') or {}).get('ms_per_step'))
"
  done
done
