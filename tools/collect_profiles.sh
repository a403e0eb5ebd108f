#!/bin/bash
# Copy what tools/profile_all.sh left under gpurun_out/ (merged back by gpurun) into profiles/ with the round's prefix:
#   bash tools/collect_profiles.sh r03
# profiles/ is tracked; gpurun_out/ is scratch.
set -eu
P=${1:?round prefix, e.g. r03}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
S=gpurun_out/profile_round/summary; E=gpurun_out/profile_extra/summary
cp $S/bench.json profiles/${P}_bench.json
cp $S/bench_trace_pipelined_kernel_stats.csv profiles/${P}_bench_trace_pipelined_kernel_stats.csv
cp $S/bench_trace_sync_kernel_stats.csv profiles/${P}_bench_trace_sync_kernel_stats.csv
cp $S/pmc_accum_l0.json profiles/${P}_pmc_accum_bpl.json
cp gpurun_out/bench_configs.jsonl profiles/${P}_bench_configs.jsonl
cp $E/vec_kernel_stats.csv profiles/${P}_vec_kernel_stats.csv
cp $E/pmc_vec_kernels.json profiles/${P}_pmc_vec_kernels.json
cp $E/vec_bench_under_rocprof.jsonl profiles/${P}_vec_bench_under_rocprof.jsonl
cp $E/bench_bls12_381.json profiles/${P}_bench_bls12_381.json
cp $E/bench_bls12_381_kernel_stats.csv profiles/${P}_bench_bls12_381_kernel_stats.csv
cp $E/profile_as_ipa_pc_as_2p16_kernel_stats.csv profiles/${P}_profile_as_ipa_pc_as_2p16_kernel_stats.csv
cp $E/profile_as_ipa_pc_as_2p16.jsonl profiles/${P}_profile_as_ipa_pc_as_2p16.jsonl
cp gpurun_out/small16/chunked_kernel_stats.csv profiles/${P}_small16_chunked_kernel_stats.csv
cp gpurun_out/small16/bps_kernel_stats.csv profiles/${P}_small16_bucket_split_kernel_stats.csv
if [ -s gpurun_out/small16/direct12_kernel_stats.csv ]; then cp gpurun_out/small16/direct12_kernel_stats.csv profiles/${P}_small12_direct_sum_kernel_stats.csv; fi
cp gpurun_out/mid_sizes.log profiles/${P}_mid_sizes.txt
(for f in r1cs_nark_as_18_harness ipa_pc_as_16_n2 hp_as_22_harness ipa_pc_as_bls_20_n2; do
   echo "== $f  (rocprofv3 --kernel-trace of build/profile_as, whole run incl. set-up; tools/trace_busy.py)"
   grep -o '"prove_ms": [0-9.]*, "verify_ms": [0-9.]*, "decide_ms": [0-9.]*' gpurun_out/scheme_trace/$f.log || true
   python3 tools/trace_busy.py gpurun_out/scheme_trace/${f}_kernel_trace.csv 0.5 0.95
   echo
 done) > profiles/${P}_scheme_trace_busy.txt
# a bench line taken AFTER the PMC summary was committed carries "sources UNCHANGED": tools/profile_all.sh's own is older
if [ -s gpurun_out/bench_final.json ]; then tail -1 gpurun_out/bench_final.json > profiles/${P}_bench.json; fi
ls -la profiles/${P}_* | wc -l
