#!/bin/bash
# rocprofv3 kernel traces of whole proves through the C++ harness (one trace per scheme / shape): where a prove's wall time
# goes -- GPU busy (union of dispatch intervals) against the gaps the host leaves.  Analysis: tools/trace_busy.py.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/scheme_trace
mkdir -p $OUT
cd /tmp
run() {  # name scheme log shape [extra]
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -- $R/build/profile_as "$@" --reps 6 --sponge poseidon --no-roundtrip > $OUT/$name.log 2>&1
  cp $(find $OUT/$name -name "*kernel_trace.csv" | head -1) $OUT/${name}_kernel_trace.csv
  rm -rf $OUT/$name
  grep '^{' $OUT/$name.log | cut -c1-600
}
run r1cs_nark_as_18_harness r1cs_nark_as 18 18 --shape harness
run ipa_pc_as_16_n2 ipa_pc_as 16 16 --shape n2
run hp_as_22_harness hp_as 22 22 --shape harness
run ipa_pc_as_bls_20_n2 ipa_pc_as 20 20 --shape n2 --curve 1
