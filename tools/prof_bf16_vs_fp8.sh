set -e
R=$(pwd)
OUT=$R/gpurun_out/r4pf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in bf16 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$d -- python $R/bench.py --dtype $d --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing --steps 80 > $OUT/$d.json 2> $OUT/$d.log
  cp "$(ls $OUT/$d/*/*kernel_stats.csv | head -1)" $OUT/${d}_kernel_stats.csv
  rm -rf $OUT/$d
  echo "$d $(grep -o '"ms_per_step": [0-9.]*' $OUT/$d.json)"
done
