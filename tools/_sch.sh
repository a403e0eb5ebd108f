mkdir -p gpurun_out/r4s
(for cfg in "AMSM_DIRECT_SUM_MAX_LOG2=16" "AMSM_DIRECT_SUM_MAX_LOG2=14"; do
echo "== $cfg"
env $cfg build/profile_as ipa_pc_as 15 16 --reps 5 --sponge poseidon --shape n2 2>&1 | grep "^JSON" | cut -c1-330
env $cfg build/profile_as ipa_pc_as 16 16 --reps 5 --sponge poseidon --shape n2 --curve 1 2>&1 | grep "^JSON" | cut -c1-330
env $cfg python tools/r4_check.py --sizes 15,16 --curves pallas,bls --kinds precomp --reps 40 2>&1 | grep batch | cut -c1-150
done
) > gpurun_out/r4s/ipa16.log 2>&1
cat gpurun_out/r4s/ipa16.log
