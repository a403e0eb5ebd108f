mkdir -p gpurun_out/r4p
for cfg in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2"; do echo "== $cfg"; env $cfg AMSM_HOST_AHEAD=1 python tools/_host.py 2>&1 | grep "ms per MSM"; env $cfg python tools/r4_check.py --no-check --sizes 20 --curves pallas --kinds precomp 2>&1 | grep batch | cut -c1-150; done > gpurun_out/r4p/q.log 2>&1
cat gpurun_out/r4p/q.log
