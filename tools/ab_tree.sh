#!/bin/bash
# same-box A/B of the working tree against another tree unpacked in _base/ (python AND library may differ):
#   rm -rf _base && mkdir _base && git archive HEAD | tar -x -C _base && make -C _base/x-ggm_amd/csrc
#   gpurun -- 'tools/ab_tree.sh 3'        (_base/ is git-ignored but travels with the snapshot; remove it afterwards)
N=${1:-3}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for i in $(seq $N); do
  for v in base new; do
    if [ $v = base ]; then D="$ROOT/_base"; else D="$ROOT"; fi
    (cd $D && python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null) | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['ms_per_pass'])
"
  done
done
