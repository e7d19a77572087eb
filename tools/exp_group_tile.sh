# per-shape GEMM times (eager HIP events, XGGM_DUMP_GEMMS) with the grouped tile pinned: which launches want which tile
mkdir -p gpurun_out/r3c
for t in 0 1 2 4; do
  XGGM_DUMP_GEMMS=1 XGGM_GROUP_TILE=$t python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loader --no-ref-batch > /dev/null 2> gpurun_out/r3c/dump_tile$t.txt
done
grep -c gemm gpurun_out/r3c/dump_tile0.txt
