mkdir -p gpurun_out/r4h
(python tools/_host.py 2>&1 | grep -v amdgpu.ids | head -6; python -m pytest tests/test_host_batch_gpu.py tests/test_direct_sum_gpu.py -q -m gpu -x 2>&1 | tail -3) > gpurun_out/r4h/host3.log 2>&1
cat gpurun_out/r4h/host3.log
