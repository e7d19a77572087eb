#!/bin/bash
# register / scratch usage of every kernel in one .hip file (product flags): tools/kernel_regs.sh gemm.hip
set -e
cd "$(dirname "$0")/../x-ggm_amd/csrc"
mkdir -p /tmp/co
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast $2 --cuda-device-only -c "$1" -o /tmp/co/dev.co 2>&1 | grep -v hip-link | head -30
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/co/dev.co --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=/tmp/co/p.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/co/p.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed|vgpr_spill" | python3 -c "
import sys
ag=None
for l in sys.stdin:
    l=l.strip()
    if 'agpr_count' in l: ag=l.split()[-1]
    elif '.name:' in l: nm=l.split()[-1].replace('_ZN12_GLOBAL__N_1','')
    elif 'private_segment' in l: sc=l.split()[-1]
    elif 'vgpr_count' in l: tot=l.split()[-1]
    elif 'vgpr_spill' in l: print(nm[:70], 'total', tot, 'agpr', ag, 'scratch', sc, 'spill', l.split()[-1])
"
