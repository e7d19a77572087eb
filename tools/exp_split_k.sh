# split-K of the FFN output product (fp32 partial slabs summed by the LayerNorm behind it): slabs against rows per GPU.
# Run ON the GPU box:   bash tools/exp_split_k.sh 92   (batch)
B=${1:-92}
for r in 1 2; do
for s in 3 0 2; do
  export XGGM_SPLIT_K=$s
  echo "batch $B split-K $s: $(python bench.py --batch $B --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
done
