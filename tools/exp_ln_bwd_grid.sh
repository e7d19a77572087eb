# LayerNorm backward: workgroups (= partial rows the batched reduction reads) per launch -- C4 (4096 rows per launch), the
# reference batch of 92 (3312 + 1840 rows) and BASELINE's batch of 32 (1152 + 640).  Run ON the GPU box:
#     bash tools/exp_ln_bwd_grid.sh 512 256 128 ...
for cap in "$@"; do
  export XGGM_LN_BWD_GRID=$cap
  a=$(python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(python bench.py --batch 92 --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  c=$(python bench.py --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "cap $cap: c4 $a   batch 92 $b   batch 32 $c"
done
