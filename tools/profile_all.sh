#!/bin/bash
# The whole evidence set of a round in one gpurun call: tools/profile_round.sh (bench line, kernel stats, PMC passes for the
# dominant kernel), tools/profile_extra.sh (vector kernels, BLS12-381, one ipa_pc_as prove), tools/bench_configs.py (the other
# BASELINE configs), tools/scheme_trace.sh (whole proves: GPU busy against host gaps), tools/small_trace.sh (a blocking 2^16 call).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
bash tools/profile_extra.sh > gpurun_out/profile_extra.log 2>&1
python3 tools/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err
bash tools/scheme_trace.sh > gpurun_out/scheme_trace.log 2>&1
bash tools/small_trace.sh > gpurun_out/small_trace.log 2>&1
python3 tools/mid_sizes.py > gpurun_out/mid_sizes.log 2>&1
tail -c 400 gpurun_out/profile_round.log
