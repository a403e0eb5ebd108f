#!/bin/bash
# rocprofv3 kernel stats of tools/r4_check.py for one configuration: tools/r4_prof.sh <tag> <r4_check args...>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/r4prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/r4_check.py --no-check "$@" > $OUT/run.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
tail -3 $OUT/run.log
column -s, -t < $OUT/kernel_stats.csv | cut -c1-60,200-330 | head -16
