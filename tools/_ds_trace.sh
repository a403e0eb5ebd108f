set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4d
mkdir -p $OUT
cd /tmp
for lg in 12 14; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ds$lg -- python3 $R/bench.py --log2n $lg --steps 40 --warmup 5 --no-cpu-baseline --no-schemes --sync --no-preheat > $OUT/ds$lg.log 2>&1
cp $(find $OUT/ds$lg -name "*kernel_stats.csv" | head -1) $OUT/ds${lg}_kernel_stats.csv; cp $(find $OUT/ds$lg -name "*kernel_trace.csv" | head -1) $OUT/ds${lg}_kernel_trace.csv; rm -rf $OUT/ds$lg
grep '^{"metric"' $OUT/ds$lg.log | tail -1 | cut -c1-300
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/ds${lg}_kernel_stats.csv")))
for r in rows[:8]:
    print(r["Name"].split("(")[0][-50:], r["Calls"], round(float(r["AverageNs"])/1e3,1),"us")
PY
done
