set -e
R=$(pwd)
OUT=$R/gpurun_out/r4pp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  if [ $v = off ]; then export XGGM_PREFETCH=0; else unset XGGM_PREFETCH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python $R/bench.py --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing --steps 60 > $OUT/$v.json 2> $OUT/$v.log
  cp "$(ls $OUT/$v/*/*kernel_stats.csv | head -1)" $OUT/${v}_kernel_stats.csv
  rm -rf $OUT/$v
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/$v.json)"
done
