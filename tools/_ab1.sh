mkdir -p gpurun_out/r4g
for cfg in "AMSM_NARROW=1" "AMSM_NARROW=0" "AMSM_NARROW=1 AMSM_BPL_PLAIN=0"; do
  echo "== $cfg"
  env $cfg build/profile_as ipa_pc_as 20 20 --shape n2 --reps 3 --sponge poseidon --curve 0 --no-roundtrip 2>&1 | grep -v amdgpu.ids | cut -c1-400
  env $cfg build/profile_as ipa_pc_as 20 20 --shape n2 --reps 3 --sponge poseidon --curve 1 --no-roundtrip 2>&1 | grep -v amdgpu.ids | cut -c1-400
  env $cfg build/profile_as ipa_pc_as 16 16 --shape n2 --reps 5 --sponge poseidon --curve 0 --no-roundtrip 2>&1 | grep -v amdgpu.ids | cut -c1-400
done > gpurun_out/r4g/ab.log 2>&1
cat gpurun_out/r4g/ab.log
