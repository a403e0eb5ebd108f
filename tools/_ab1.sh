mkdir -p gpurun_out/r4u
for i in 1 2; do
for cfg in "AMSM_NARROW=1" "AMSM_NARROW=0 AMSM_RED2=0"; do
  v=$(env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schemes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), d['config']['ms_per_msm_synchronous_call'])")
  echo "$cfg: $v"
done; done > gpurun_out/r4u/b.log 2>&1
cat gpurun_out/r4u/b.log
