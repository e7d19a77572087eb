set -e
R=$(pwd)
OUT=$R/gpurun_out/r4dp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in plain force; do
  if [ $v = force ]; then export XGGM_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29578; else unset XGGM_DP_FORCE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python $R/bench.py --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing --steps 60 > $OUT/$v.json 2> $OUT/$v.log
  cp "$(ls $OUT/$v/*/*kernel_stats.csv | head -1)" $OUT/${v}_kernel_stats.csv
  rm -rf $OUT/$v
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/$v.json)"
done
