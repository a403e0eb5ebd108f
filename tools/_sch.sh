mkdir -p gpurun_out/r4s
(python -m pytest tests/test_direct_sum_gpu.py tests/test_two_valued_gpu.py tests/test_msm_gpu.py -q -m gpu -x 2>&1 | tail -4
for cfg in "AMSM_DIRECT_SUM_MAX_LOG2=14"; do
  echo "== $cfg"
  env $cfg build/profile_as all 10 10 --reps 7 --sponge poseidon --shape both 2>&1 | grep "^JSON" | cut -c1-330
  env $cfg build/profile_as all 12 12 --reps 7 --sponge poseidon --shape n2 2>&1 | grep "^JSON" | cut -c1-330
  env $cfg build/profile_as all 14 14 --reps 7 --sponge poseidon --shape n2 2>&1 | grep "^JSON" | cut -c1-330
done
python tools/r4_check.py --no-check --sizes 10,12,14 --curves pallas,bls --kinds precomp --reps 40 2>&1 | grep batch | cut -c1-150
) > gpurun_out/r4s/schemes2.log 2>&1
cat gpurun_out/r4s/schemes2.log
