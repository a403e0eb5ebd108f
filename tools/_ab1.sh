mkdir -p gpurun_out/r4q
for cfg in "AMSM_RED2=1" "AMSM_RED2=0"; do
  echo "== $cfg"
  env $cfg python tools/r4_check.py --sizes 12,14,16,17,18,19,20 --curves pallas --kinds precomp 2>&1 | grep batch | cut -c1-150
  env $cfg python tools/r4_check.py --sizes 18,20 --curves pallas --kinds plain 2>&1 | grep batch | cut -c1-150
  env $cfg python tools/r4_check.py --sizes 18,20 --curves bls --kinds precomp 2>&1 | grep batch | cut -c1-150
done > gpurun_out/r4q/ab.log 2>&1
cat gpurun_out/r4q/ab.log
