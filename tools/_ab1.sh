mkdir -p gpurun_out/r4l
for cfg in "AMSM_BPS_MAX_LOG2=17" "AMSM_BPS_MAX_LOG2=19" "AMSM_BPS_MAX_LOG2=19 AMSM_BPS_WANT=256" "AMSM_BPS_MAX_LOG2=19 AMSM_BPS_WANT=64"; do
  echo "== $cfg"
  env $cfg python tools/r4_check.py --no-check --sizes 18,19 --curves pallas,bls --kinds precomp 2>&1 | grep -v amdgpu.ids | cut -c1-220
done > gpurun_out/r4l/ab.log 2>&1
cat gpurun_out/r4l/ab.log
