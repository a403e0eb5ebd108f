mkdir -p gpurun_out/r4w
(for cfg in "AMSM_RADIX=1" "AMSM_RADIX=0"; do
  echo "== $cfg"
  env $cfg python tools/r4_check.py --sizes 18,19,20 --curves pallas,bls --kinds precomp 2>&1 | grep batch | cut -c1-175
done
echo "== tests under AMSM_RADIX=1"
AMSM_RADIX=1 python -m pytest tests/test_narrow_gpu.py tests/test_fold_gpu.py tests/test_bpl_gpu.py -q -m gpu -x -k "grouped or fold or uniform or sizes_inside or ranges or non_canonical or bls12_381_at or windows_of" 2>&1 | tail -15) > gpurun_out/r4w/radix.log 2>&1
cat gpurun_out/r4w/radix.log
