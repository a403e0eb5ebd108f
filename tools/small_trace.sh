set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/small16
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chunked -- python3 $R/bench.py --log2n 16 --steps 40 --warmup 5 --no-cpu-baseline --no-schemes --sync > $OUT/chunked.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bps -- python3 $R/bench.py --log2n 16 --steps 40 --warmup 5 --no-cpu-baseline --no-schemes --sync > $OUT/bps.log 2>&1
# round 4: a blocking 2^12-pair MSM per step -- the direct sum over a small key (k_direct_sum + k_fold_quad)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/direct12 -- python3 $R/bench.py --log2n 12 --steps 40 --warmup 5 --no-cpu-baseline --no-schemes --sync --no-preheat > $OUT/direct12.log 2>&1
for d in chunked bps direct12; do cp $(find $OUT/$d -name "*kernel_stats.csv" | head -1) $OUT/${d}_kernel_stats.csv; cp $(find $OUT/$d -name "*kernel_trace.csv" | head -1) $OUT/${d}_kernel_trace.csv; rm -rf $OUT/$d; grep '^{"metric"' $OUT/$d.log | tail -1 | cut -c1-400; done
