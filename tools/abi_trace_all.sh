#!/bin/bash
# tools/abi_trace.py over the config-size accumulations (the last timed prove of each): where the host time goes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for spec in "r1cs_nark_as 18 harness" "r1cs_nark_as 18 n2" "ipa_pc_as 16 n2" "ipa_pc_as 16 harness" "hp_as 22 n2" "hp_as 22 harness" "trivial_pc_as 10 harness"; do
  set -- $spec
  echo "==== $1 2^$2 $3 (poseidon)"
  python3 tools/abi_trace.py -- build/profile_as $1 $2 $2 --shape $3 --sponge poseidon --reps 3 --no-roundtrip 2>&1 | grep -v "^$" | grep -A40 "^JSON" | cut -c1-400
done
