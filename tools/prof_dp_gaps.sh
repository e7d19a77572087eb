# where the one-rank run of the N > 1 path loses its time: kernel-trace gaps of the plain step against the forced
# one-rank RCCL group.  Run ON the GPU box:  bash tools/prof_dp_gaps.sh
R=$(pwd); OUT=$R/gpurun_out/dpgaps; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in plain force; do
  if [ $v = force ]; then export XGGM_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29579; else unset XGGM_DP_FORCE; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -- python $R/bench.py --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing --steps 8 --warmup 2 > $OUT/$v.json 2> $OUT/$v.log
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $OUT/$v.json)"
  python $R/tools/trace_gaps.py $OUT/$v 4 14
  rm -rf $OUT/$v
done
