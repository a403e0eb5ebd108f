mkdir -p gpurun_out/r4r
for t in 0 12 13 14 15 16 17 18 99; do
  for cv in 0 1; do
    r=$(AMSM_IPA_FOLD_ABOVE=$t build/profile_as ipa_pc_as 20 20 --shape n2 --reps 3 --sponge poseidon --curve $cv --no-roundtrip 2>&1 | grep -o '"prove_ms": [0-9.]*')
    echo "fold_above=$t curve=$cv 2^20 $r"
  done
  r=$(AMSM_IPA_FOLD_ABOVE=$t build/profile_as ipa_pc_as 16 16 --shape n2 --reps 5 --sponge poseidon --curve 0 --no-roundtrip 2>&1 | grep -o '"prove_ms": [0-9.]*')
  echo "fold_above=$t curve=0 2^16 $r"
  r=$(AMSM_IPA_FOLD_ABOVE=$t build/profile_as ipa_pc_as 18 18 --shape n2 --reps 3 --sponge poseidon --curve 0 --no-roundtrip 2>&1 | grep -o '"prove_ms": [0-9.]*')
  echo "fold_above=$t curve=0 2^18 $r"
done > gpurun_out/r4r/fold.log 2>&1
cat gpurun_out/r4r/fold.log
