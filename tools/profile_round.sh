#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py (pipelined and --sync) and PMC passes for the dominant kernel.
# Counters are collected in their own runs (never combined with sys/hip tracing), one --pmc group per pass.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_round
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python3 $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-schemes > $OUT/trace_pipelined.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sync -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-schemes --sync > $OUT/trace_sync.log 2>&1
for p in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $p | cut -c1-10 | tr " " "_")
  rocprofv3 --pmc $p --kernel-include-regex "k_accum_(l0|bpl)" --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-schemes --sync > $OUT/pmc_$n.log 2>&1
done
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
# summaries to copy into profiles/ (tracked)
mkdir -p $OUT/summary
cp $(find $OUT/trace_pipelined -name "*kernel_stats.csv" | head -1) $OUT/summary/bench_trace_pipelined_kernel_stats.csv
cp $(find $OUT/trace_sync -name "*kernel_stats.csv" | head -1) $OUT/summary/bench_trace_sync_kernel_stats.csv
python3 $R/tools/pmc_summary.py $OUT $OUT/summary/pmc_accum_l0.json
cp $OUT/bench.json $OUT/summary/bench.json
tail -c 600 $OUT/bench.json
