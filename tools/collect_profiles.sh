#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU call:  bash tools/collect_profiles.sh gpurun_out/r2p
# (kernel-trace statistics of the bench commands; HBM-side traffic and MFMA occupancy in separate --pmc passes
# over eager launches, as MI355X_MICROARCH.md prescribes)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench flags
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python "$R/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.log"
    cp "$(ls "$OUT/$name"/*/*kernel_stats.csv | head -1)" "$OUT/${name}_kernel_stats.csv"
    rm -f "$OUT/$name"/*/*kernel_trace.csv
    echo "$name: $(tail -1 "$OUT/$name.json" | cut -c1-40) ... $(tail -1 "$OUT/$name.json" | grep -o '"ms_per_step": [0-9.]*')"
}
if [ -z "$PMC_ONLY" ]; then python "$R/bench.py" > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"; fi
[ -n "$PMC_ONLY" ] || echo "plain: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_plain.json" | head -1)"
# (--no-ref-batch: the leg at the reference's batch of 92 / 96 runs the same kernels on three times the rows and would
# mix into the per-kernel averages; the un-profiled line above carries it)
if [ -z "$PMC_ONLY" ]; then
stats bench --no-ref-batch
stats bench_fp8 --dtype fp8 --no-cpu-baseline --no-ref-batch
stats bench_c4 --workload c4 --no-cpu-baseline
stats bench_gqa --order gqa --no-cpu-baseline --no-ref-batch
fi
if [ -n "$STATS_ONLY" ]; then echo done; exit 0; fi
EAGER="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-loader --no-ref-batch"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python "$R/bench.py" $EAGER > "$OUT/pmc_$c.log" 2>&1
    cp "$(ls "$OUT/pmc_$c"/*/*counter_collection.csv | head -1)" "$OUT/pmc_$c.csv"
    echo "pmc $c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python "$R/bench.py" $EAGER > "$OUT/pmc_mfma.log" 2>&1
python "$R/tools/pmc_summary.py" "$OUT/pmc_FETCH_SIZE.csv" "$OUT/pmc_WRITE_SIZE.csv" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt"
python "$R/tools/pmc_mfma.py" "$OUT/pmc_mfma" > "$OUT/pmc_mfma.txt"
rm -rf "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_mfma "$OUT"/pmc_*.csv
echo done
